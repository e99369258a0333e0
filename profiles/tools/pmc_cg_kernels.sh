#!/bin/bash
# Counters of the kernels of a batched CG iteration (8 images, 256 x 256, gaussian_blur + inpainting; m = 0 or 16):
# MFMA-pipe busy fraction, held clock, LDS activity / bank conflicts and L2-side traffic of k_dct_sym and the vector kernels.
#   bash profiles/tools/pmc_cg_kernels.sh [nsteps: 0 -> m = 0, 8 -> m = 16]     (separate --pmc passes, counters only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NS=${1:-0}
O=gpurun_out/pmc_cg; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 profiles/tools/prof_cgb.py 8 $NS > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- python3 profiles/tools/prof_cgb.py 8 $NS > $O/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/p3 -- python3 profiles/tools/prof_cgb.py 8 $NS > $O/p3.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS --output-format csv -d $O/p4 -- python3 profiles/tools/prof_cgb.py 8 $NS > $O/p4.log 2>&1
python3 - <<PY
import csv, glob, json, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_cg/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_dct_sym<[a-z]*>|k_cg_step1|k_cg_step2|k_rep_dots<[a-z]*>|k_rep_apply2<[a-z]*>|k_rep_coef|k_conv1d<[0-9]>)", row["Kernel_Name"])
        if m:
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[m.group(1)]["duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
out = {"what": "kernels of profiles/tools/prof_cgb.py 8 $NS (batched CG, 8 images, 256 x 256, 64 iterations per operator), MI355X, rocprofv3 --pmc (four passes); FETCH_SIZE / WRITE_SIZE in KiB per the guide (FETCH_SIZE x 2 on gfx950)"}
for k, d in sorted(acc.items()):
    rec = {c: round(sum(v) / len(v)) for c, v in d.items()}
    rec["launches_counted"] = len(d["duration_ns"])
    der = {}
    if "GRBM_GUI_ACTIVE" in rec and rec["duration_ns"]:
        cyc = rec["GRBM_GUI_ACTIVE"] / 8
        der["effective_clock_GHz"] = round(cyc / rec["duration_ns"], 2)
        der["mfma_pipe_busy_fraction_at_held_clock"] = round(rec.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 3)
    if "FETCH_SIZE" in rec:
        der["l2_side_read_MB"] = round(rec["FETCH_SIZE"] * 2 * 1024 / 1e6, 1)
        der["l2_side_write_MB"] = round(rec.get("WRITE_SIZE", 0) * 1024 / 1e6, 1)
    rec["derived"] = der
    out[k] = rec
json.dump(out, open("gpurun_out/r03_cg_kernels_pmc_ns$NS.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
