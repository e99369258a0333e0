#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trace_cg; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 profiles/tools/prof_cgb.py 8 0 > $O/run.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob("gpurun_out/trace_cg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# classify dct passes by the preceding / following kernel
stat=collections.defaultdict(list); gaps=[]
prev=None
for i,(s,e,n) in enumerate(rows):
    short = "dct_inv" if "k_dct_sym<true" in n else "dct_fwd" if "k_dct_sym<false" in n else n.split("(")[0][:24]
    if short.startswith("dct"):
        pn = rows[i-1][2] if i>0 else ""
        first = not ("k_dct_sym" in pn and (("<true" in pn) == ("<true" in n)))
        short += "_p1" if first else "_p2"
    stat[short].append((e-s)/1e3)
    if prev is not None: gaps.append((s-prev)/1e3)
    prev=e
for k,v in sorted(stat.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v); print("%-28s n=%6d mean %7.2f us  median %7.2f  p10 %7.2f p90 %7.2f" % (k,len(v),sum(v)/len(v),v2[len(v2)//2],v2[len(v2)//10],v2[9*len(v2)//10]))
g=sorted(gaps); print("gaps between consecutive kernels: median %.2f us, mean %.2f us, p90 %.2f" % (g[len(g)//2], sum(g)/len(g), g[9*len(g)//10]))
PY
