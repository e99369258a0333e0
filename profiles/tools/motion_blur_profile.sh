cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FH_PHASE_TIMES=1 python3 bench.py --operator motion_blur --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/mb_phase.log 2>&1
grep "FH_PHASE_TIMES" gpurun_out/mb_phase.log | tail -1
rm -rf gpurun_out/prof_mb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mb -- python3 bench.py --operator motion_blur --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/mb_prof.log 2>&1
python3 - <<PY
import glob,csv
f=glob.glob("gpurun_out/prof_mb/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%6.2f%% %9d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:90]))
print("total kernel ms", tot/1e6)
PY
rm -rf gpurun_out/prof_mb
