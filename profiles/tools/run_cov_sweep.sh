python3 -m pytest tests/test_hip_parity256.py -q -m gpu -k "single_sweep" -x > gpurun_out/t_fused.log 2>&1; tail -3 gpurun_out/t_fused.log
for st in 0; do python3 profiles/tools/prof_cov_fused.py 8 32 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['us_per_apply'], d['frac'], d['other_variant']['us_per_apply'])"; done
echo b1; python3 profiles/tools/prof_cov_fused.py 1 32 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['us_per_apply'], d['frac'], d['other_variant']['us_per_apply'])"
echo b8m16; python3 profiles/tools/prof_cov_fused.py 8 16 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['us_per_apply'], d['frac'], d['other_variant']['us_per_apply'])"
echo b8m56; python3 profiles/tools/prof_cov_fused.py 8 56 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['us_per_apply'], d['frac'], d['other_variant']['us_per_apply'])"
