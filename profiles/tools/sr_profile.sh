#!/bin/bash
# rocprofv3 kernel stats of the SR x4 workload (FFHQ arch, batch 8, N = 1): the operator kernels k_conv_dec / k_conv_up
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_sr
FH_PHASE_TIMES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sr -- python3 bench.py --operator super_resolution --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/sr_prof.log 2>&1
grep "FH_PHASE_TIMES" gpurun_out/sr_prof.log | tail -1
python3 - <<PY
import glob,csv
f=glob.glob("gpurun_out/prof_sr/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if any(k in r["Name"] for k in ("k_conv_dec","k_conv_up","k_conv_tile","k_conv_direct","k_dct_sym","k_rep_","k_cg_","k_dot")):
        print("%6.2f%% %9d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:80]))
print("total kernel ms", tot/1e6)
PY
rm -rf gpurun_out/prof_sr
