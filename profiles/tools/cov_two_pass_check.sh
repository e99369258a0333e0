#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_hip_parity256.py tests/test_hip_parity.py tests/test_batched_sampler.py -q -m gpu -k "single_sweep or apply or batched or covariance or solver" -x > gpurun_out/t_cov2.log 2>&1; tail -3 gpurun_out/t_cov2.log
for cfg in "8 32" "1 32" "8 16" "8 56" "8 8"; do echo "b m = $cfg"; python3 profiles/tools/prof_cov_fused.py $cfg 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['kernel'][:40], d['us_per_apply'], d['frac'], d['other_variant']['us_per_apply'])"; done
rm -rf gpurun_out/prof_cov2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cov2 -- python3 profiles/tools/prof_cov_one.py 8 0 32 > /dev/null 2>&1
cut -c1-160 gpurun_out/prof_cov2/*/*kernel_stats.csv | head -5
