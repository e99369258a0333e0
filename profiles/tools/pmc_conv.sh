#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE in separate --pmc passes) of the UNet convolution of bench.py's roofline_unet_conv
# (3x3 128 -> 128 on 8 x 256 x 256 NHWC): split-bf16 kernel and fp32-MFMA kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 profiles/tools/prof_conv_x6.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 profiles/tools/prof_conv_x6.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 profiles/tools/prof_conv_x6.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, re, collections
rec = {"what": "UNet convolution 3x3 128->128 on 8x256x256 NHWC float32 (bench.py roofline_unet_conv), MI355X, rocprofv3 7.2",
       "commands": ["bash profiles/tools/pmc_conv.sh  (rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE as SEPARATE passes of python3 profiles/tools/prof_conv_x6.py)"],
       "units": "counter values are KiB per dispatch (mean over the dispatches of the run); FETCH_SIZE doubled for gfx950 wide coalesced reads (MI355X_MICROARCH.md, HBM section)",
       "algorithmic_bytes": {"input": 8 * 256 * 256 * 128 * 4, "output": 8 * 256 * 256 * 128 * 4, "weights_bf16_x3": 9 * 128 * 128 * 2 * 3}}
f = glob.glob("gpurun_out/pmc_conv/stats/**/*kernel_stats.csv", recursive=True)[0]
us = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_conv_[a-z0-9_]+<[^>]*>)", r["Name"])
    if m: us[m.group(1)] = round(float(r["AverageNs"]) / 1e3, 1)
rec["kernels_us"] = us
for kind, key in (("fetch", "FETCH_SIZE_KiB"), ("write", "WRITE_SIZE_KiB")):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_conv/{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_conv_[a-z0-9_]+<[^>]*>)", r["Kernel_Name"])
            if m: acc[m.group(1)].append(float(r["Counter_Value"]))
    rec[key] = {k: round(sum(v) / len(v), 1) for k, v in acc.items()}
rec["traffic_bytes_per_launch"] = {k: int(rec["FETCH_SIZE_KiB"][k] * 2048 + rec["WRITE_SIZE_KiB"].get(k, 0) * 1024) for k in rec["FETCH_SIZE_KiB"]}
algo = sum(rec["algorithmic_bytes"].values())
rec["traffic_over_algorithmic"] = {k: round(v / algo, 3) for k, v in rec["traffic_bytes_per_launch"].items()}
json.dump(rec, open("gpurun_out/r03_conv_pmc.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
rm -rf $O
