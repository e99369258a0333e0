#!/usr/bin/env python3
"""Reference-vs-reference spread of the free-running trajectories: the ORACLE (the reference's arithmetic restated on the
CPU, pinned to the reference by tests/test_oracle_golden.py) run on THIS host's CPU against the recordings the reference
produced on the build container's CPU (tests/golden/trajectories.npz).  Same code, same inputs, different CPU: whatever
differs here is what "identical seeds" cannot pin for the rounding-chaotic configurations, and is the yardstick the
free-running HIP tests (tests/test_hip_parity256.py) are held to.

    python3 profiles/tools/oracle_spread.py > gpurun_out/r02_free_running_spread.json
"""
import json, os, platform, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
from test_oracle_golden import T, run_oracle_traj
from oracle import fh_oracle as fo

g = np.load(os.path.join(ROOT, "tests", "golden", "trajectories.npz"), allow_pickle=False)
sg = np.load(os.path.join(ROOT, "tests", "golden", "sigma_grids.npz"), allow_pickle=False)
tmp = tempfile.mkdtemp()
torch.save(T(g["dct_variance64"]), os.path.join(tmp, "dct_variance.pt"))
if (os.cpu_count() or 1) > 32:
    torch.set_num_threads(32)
cpu = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), platform.processor())
out = {"host_cpu": cpu, "threads": torch.get_num_threads(),
       "sigma_table_bit_identical_to_recording_host": bool(np.array_equal(fo.linear_sigma_table().numpy(), sg["u"]))}
for tag in ["gb_heun10", "mb_heun10", "ip_euler20", "gb_heun30", "sr_heun10", "gb_heun10_identity"]:
    if tag == "gb_heun30" and "--quick" in sys.argv:
        continue
    x, mech = run_oracle_traj(g, tag, tmp)
    p = tag + "__"
    tr = mech.trace
    n_o, n_r = np.array([t["niter"] for t in tr]), np.asarray(g[p + "niter"])
    b_o, b_r = np.array([int(t["branch"] == "cov") for t in tr]), np.asarray(g[p + "branch_cov"])
    ref = T(g[p + "x_final"]).double()
    diff = x.double() - ref
    mse = float((diff ** 2).mean())
    out[tag] = {"calls": len(tr), "k_equal": [t["k"] for t in tr] == list(g[p + "k"]),
                "branch_mismatch_calls": int((b_o != b_r).sum()), "niter_equal_calls": int((n_o == n_r).sum()),
                "niter_sum_oracle_here": int(n_o.sum()), "niter_sum_ref": int(n_r.sum()),
                "niter_max_rel_dev": float((np.abs(n_o - n_r) / np.maximum(n_r, 1)).max()),
                "final_max_abs": float(diff.abs().max()), "final_rms": float(mse ** 0.5),
                "final_psnr_vs_ref_db": 99.0 if mse == 0 else float(10 * np.log10(4.0 / mse))}
if "--256" in sys.argv:  # full-size recordings (sr: 813 CG iterations, the cheapest; gb: the headline operator)
    import inputs
    g256 = np.load(os.path.join(ROOT, "tests", "golden", "trajectories256.npz"), allow_pickle=False)
    data = os.path.join(ROOT, "free-hunch_amd", "data")
    for tag in ["sr256_heun30", "gb256_heun30"]:
        x, mech = run_oracle_traj(g256, tag, data, size=256, cfg=inputs.SMALL_C)
        p = tag + "__"
        tr = mech.trace
        n_o, n_r = np.array([t["niter"] for t in tr]), np.asarray(g256[p + "niter"])
        b_o, b_r = np.array([int(t["branch"] == "cov") for t in tr]), np.asarray(g256[p + "branch_cov"])
        diff = x.double()[..., ::4, ::4] - T(g256[p + "x_final"]).double()
        mse = float((diff ** 2).mean())
        out[tag] = {"calls": len(tr), "k_equal": [t["k"] for t in tr] == list(g256[p + "k"]),
                    "branch_mismatch_calls": int((b_o != b_r).sum()), "niter_equal_calls": int((n_o == n_r).sum()),
                    "niter_sum_oracle_here": int(n_o.sum()), "niter_sum_ref": int(n_r.sum()),
                    "niter_max_rel_dev": float((np.abs(n_o - n_r) / np.maximum(n_r, 1)).max()),
                    "final_max_abs": float(diff.abs().max()), "final_rms": float(mse ** 0.5),
                    "final_psnr_vs_ref_db": 99.0 if mse == 0 else float(10 * np.log10(4.0 / mse))}
print(json.dumps(out, indent=1))
