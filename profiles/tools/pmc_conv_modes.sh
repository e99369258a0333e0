#!/bin/bash
# MFMA-pipe busy fraction, held clock and LDS counters of the dominant 3x3 convolutions in one precision mode (default 4 = half-split):
#   bash profiles/tools/pmc_conv_modes.sh [mode]      (separate --pmc passes, counters only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MODE=${1:-4}
O=gpurun_out/pmc_conv_modes; rm -rf $O; mkdir -p $O
ITERS=5 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 profiles/tools/bench_conv_modes.py $MODE > /dev/null 2>&1
ITERS=5 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- python3 profiles/tools/bench_conv_modes.py $MODE > $O/p2.log 2>&1
python3 - <<PY
import csv, glob, json, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_conv_modes/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_conv_x6r<[^>]*>)", row["Kernel_Name"])
        if m:
            key = m.group(1) + " grid=" + row.get("Grid_Size", "?")
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[key]["duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
out = {"what": "3x3 convolutions of profiles/tools/bench_conv_modes.py, mode $MODE, MI355X, rocprofv3 --pmc (two passes)"}
for k, d in sorted(acc.items()):
    rec = {c: round(sum(v) / len(v)) for c, v in d.items()}
    if "GRBM_GUI_ACTIVE" in rec and rec["duration_ns"]:
        clk = rec["GRBM_GUI_ACTIVE"] / 8 / rec["duration_ns"]
        rec["derived"] = {"effective_clock_GHz": round(clk, 2),
                          "mfma_pipe_busy_fraction_at_held_clock": round(rec.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (rec["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)}
    out[k] = rec
json.dump(out, open("gpurun_out/r03_conv_modes_pmc_mode$MODE.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
