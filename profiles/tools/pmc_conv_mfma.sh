#!/bin/bash
# MFMA-pipe busy fraction and held clock of the UNet convolution of bench.py's roofline_unet_conv (one --pmc pass, counters only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_conv_mfma; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O -- python3 profiles/tools/prof_conv_x6.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_conv_mfma/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_conv_[a-z0-9_]+)", row["Kernel_Name"])
        if m:
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[m.group(1)]["duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
out = {"what": "UNet convolution 3x3 128->128 on 8x256x256 NHWC (bench.py roofline_unet_conv), MI355X, rocprofv3 7.2, one --pmc pass",
       "command": "bash profiles/tools/pmc_conv_mfma.sh"}
for k, d in acc.items():
    rec = {c: round(sum(v) / len(v)) for c, v in d.items()}
    rec["dispatches"] = len(d["duration_ns"])
    clk = rec["GRBM_GUI_ACTIVE"] / 8 / rec["duration_ns"]            # GUI-active cycles are summed over the 8 XCDs
    rec["derived"] = {"effective_clock_GHz": round(clk, 2),
                      "mfma_pipe_busy_fraction_at_held_clock": round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)}
    out[k] = rec
json.dump(out, open("gpurun_out/r03_conv_mfma_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
