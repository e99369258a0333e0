#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trace_bench; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-half-split-leg > $O/run.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob("gpurun_out/trace_bench/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
print("kernels", len(rows))
# segment: UNet phases = runs of kernels containing k_conv / k_gn; find windows of continuous activity (gap < 2 ms)
busy=0; gaps=collections.Counter(); gap_time=collections.Counter()
t0=rows[0][0]; prev_end=rows[0][1]
segs=[]; seg_start=rows[0][0]; seg_busy=0
for s,e,n in rows:
    g=s-prev_end
    if g>2_000_000:
        segs.append((seg_start, prev_end, seg_busy)); seg_start=s; seg_busy=0
    elif g>0:
        b = "<5us" if g<5000 else "<20us" if g<20000 else "<100us" if g<100000 else "<2ms"
        gaps[b]+=1; gap_time[b]+=g
    seg_busy+=max(0,e-max(s,prev_end)) if e>prev_end else 0
    prev_end=max(prev_end,e)
segs.append((seg_start, prev_end, seg_busy))
tot=sum(e-s for s,e,_ in segs); bz=sum(b for _,_,b in segs)
print("active segments %d, total span %.3f s, GPU busy %.3f s (%.1f%%)" % (len(segs), tot/1e9, bz/1e9, 100*bz/tot))
for k in ["<5us","<20us","<100us","<2ms"]:
    print("gaps %-7s n=%7d total %.3f s" % (k, gaps[k], gap_time[k]/1e9))
big=sorted(segs,key=lambda x:-(x[1]-x[0]))[:6]
for s,e,b in big: print("segment %.3f s busy %.1f%%" % ((e-s)/1e9, 100*b/(e-s)))
PY
