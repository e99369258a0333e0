#!/bin/bash
# phase times + rocprofv3 kernel stats of the ImageNet-256 architecture (SR x4, batch 8, N = 1); writes under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_in
FH_PHASE_TIMES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_in -- python3 bench.py --arch imagenet --operator super_resolution --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/in_prof.log 2>&1
grep "FH_PHASE_TIMES" gpurun_out/in_prof.log | tail -1
python3 - <<PY
import glob,csv,shutil
f=glob.glob("gpurun_out/prof_in/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%6.2f%% %9d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
print("total kernel ms", tot/1e6)
shutil.copy(f, "gpurun_out/r03_bench_imagenet_sr_heun30_b8_kernel_stats.csv")
PY
rm -rf gpurun_out/prof_in
