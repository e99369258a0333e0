#!/bin/bash
# per-kernel times of the batched CG (8 images, m = 16) - rocprofv3 kernel stats of profiles/tools/prof_cgb.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cgb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cgb -- python3 profiles/tools/prof_cgb.py 8 ${1:-8} > /dev/null 2>&1
python3 - <<PY
import glob,csv
f=glob.glob("gpurun_out/prof_cgb/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%6.2f%% %7d calls %8.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:80]))
PY
