#!/bin/bash
# phase times + rocprofv3 kernel stats of the default bench command (N = 1); writes under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FH_PHASE_TIMES=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_phase.log 2>&1
grep "FH_PHASE_TIMES" gpurun_out/bench_phase.log | tail -1
tail -1 gpurun_out/bench_phase.log | cut -c1-400
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
python3 - <<PY
import glob,csv
f=glob.glob("gpurun_out/prof_bench/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    print("%6.2f%% %9d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:100]))
print("total kernel ms", tot/1e6)
import shutil; shutil.copy(f, "gpurun_out/bench_kernel_stats.csv")
PY
