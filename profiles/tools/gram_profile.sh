cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 profiles/tools/bench_gram.py 2>&1 | tail -6
rm -rf gpurun_out/prof_gram
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gram -- python3 profiles/tools/bench_gram.py > /dev/null 2>&1
cut -c1-150 gpurun_out/prof_gram/*/*kernel_stats.csv | head -6
python3 -m pytest tests/test_hip_parity.py tests/test_hip_parity256.py -q -m gpu -k "covariance or extended" --timeout 300 2>&1 | tail -2
rm -rf gpurun_out/prof_gram
