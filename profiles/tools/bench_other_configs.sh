#!/bin/bash
# the other operators / architectures of BASELINE.json's configs through bench.py (N = 1, batch 8), one JSON line each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_bench_other_configs.jsonl; : > $out
run() { python3 bench.py --no-cpu-baseline --no-half-split-leg --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps({k: j[k] for k in ('metric','value','unit','ms_per_step','steps','config','dtype')}))" >> $out; tail -1 $out | cut -c1-300; }
run --operator motion_blur
run --operator super_resolution
run --operator inpainting --solver euler --num-steps 100
run --arch imagenet --operator super_resolution
run --arch imagenet --operator motion_blur
