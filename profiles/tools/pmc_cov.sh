#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE in separate --pmc passes) and kernel times of the cov-apply at d = 196608, m = 32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_cov; rm -rf $O; mkdir -p $O
for cfg in "8 0 b8_twopass" "1 2 b1_singlesweep" "1 0 b1_twopass"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$3_stats -- python3 profiles/tools/prof_cov_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$3_fetch -- python3 profiles/tools/prof_cov_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$3_write -- python3 profiles/tools/prof_cov_one.py $1 $2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, re, collections, shutil
out = {}
for tag, nimg in (("b8_twopass", 8), ("b1_singlesweep", 1), ("b1_twopass", 1)):
    rec = {"kernels_us": {}, "FETCH_SIZE_KiB": {}, "WRITE_SIZE_KiB": {}}
    f = glob.glob(f"gpurun_out/pmc_cov/{tag}_stats/**/*kernel_stats.csv", recursive=True)[0]
    shutil.copy(f, f"gpurun_out/r03_cov_apply_{tag}_kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_rep_[a-z0-9_]+)", r["Name"])
        if m: rec["kernels_us"][m.group(1)] = round(float(r["AverageNs"]) / 1e3, 2)
    for kind, key in (("fetch", "FETCH_SIZE_KiB"), ("write", "WRITE_SIZE_KiB")):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"gpurun_out/pmc_cov/{tag}_{kind}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"(k_rep_[a-z0-9_]+)", r["Kernel_Name"])
                if m: acc[m.group(1)].append(float(r["Counter_Value"]))
        rec[key] = {k: round(sum(v) / len(v), 2) for k, v in acc.items()}
    fetch = sum(rec["FETCH_SIZE_KiB"].values()) * 1024 * 2   # gfx950: FETCH_SIZE reports half of wide coalesced reads
    write = sum(rec["WRITE_SIZE_KiB"].values()) * 1024
    rec["traffic_bytes_per_apply"] = int(fetch + write)
    rec["algorithmic_bytes_per_apply"] = nimg * 56623104
    rec["traffic_over_algorithmic"] = round((fetch + write) / (nimg * 56623104), 3)
    rec["us_per_apply_kernel_sum"] = round(sum(rec["kernels_us"].values()), 2)
    out[tag] = rec
json.dump(out, open("gpurun_out/r03_cov_apply_pmc_all.json", "w"), indent=1)
print(json.dumps(out, indent=1))
# the two files bench.py reads its `traffic` from
cmd = ["bash profiles/tools/pmc_cov.sh  (rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE as SEPARATE passes of `python3 profiles/tools/prof_cov_one.py <nimg> <exclusive>`)"]
units = "counter values are KiB per dispatch (mean over 30 dispatches); FETCH_SIZE doubled for gfx950 wide coalesced reads (MI355X_MICROARCH.md, HBM section)"
b8 = {"what": "fh_rep_apply_batched at d=196608, m=32, float64, 8 images per launch: the two-pass kernels the lock-step CG issues (k_rep_dots<nt> + k_rep_coef + k_rep_apply2<nt>), MI355X, rocprofv3 7.2",
      "commands": cmd, "units": units, **out["b8_twopass"],
      "note": "1.95x the algorithmic bytes: each image's factor base is read by both passes (the counters also count Infinity-Cache hits).  The single-sweep kernel reads it once (r03_cov_apply_pmc.json) but is slower for batched launches - profiles/r02_cov_apply_single_sweep.md."}
json.dump(b8, open("gpurun_out/r03_cov_apply_b8_pmc.json", "w"), indent=1)
b1 = {"what": "fh_rep_apply at d=196608, m=32, float64, ONE image per launch on an exclusive context: the single-sweep kernel k_rep_fused<4>",
      "commands": cmd, "units": units, **out["b1_singlesweep"], "two_pass_for_comparison": out["b1_twopass"]}
json.dump(b1, open("gpurun_out/r03_cov_apply_pmc.json", "w"), indent=1)
PY
rm -rf $O
