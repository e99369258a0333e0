#!/usr/bin/env python3
"""Benchmark of the Free Hunch hot path on MI355X (contract: see the task statement / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--arch ffhq|imagenet] [--operator NAME]

One "step" = one batch of B independent 256x256 images, each taken through the full FH-Heun sampler
(num_steps = 30 -> 59 guidance calls = 59 UNet forwards + 59 input-VJPs + 59 CG solves + covariance updates).
Workload at N = 1: BASELINE.json configs[1] (FFHQ-256 architecture, gaussian_blur, low-rank covariance,
num_steps = 30, batch = 8).  N > 1: one process per GPU (torch.distributed / RCCL), every rank runs its own batch
(weak scaling), outputs are exchanged with ONE all_gather per step inside the timed region.
Inputs are synthetic (seeded images and weights; there are no checkpoints or datasets offline).

Besides the headline line the JSON carries
  roofline      cov-apply (fh_rep_apply at d = 196608, m = 32) timed with events on the launch stream,
                algorithmic bytes / time against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference path) timed on the host cores on a bounded sample of the
                same workload (the first guidance calls of one image), extrapolated to images/s
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def smooth_images(n, size, seed):
    g = np.random.default_rng(seed)
    f = g.standard_normal((n, 3, size, size))
    fy, fx = np.fft.fftfreq(size)[:, None], np.fft.fftfreq(size)[None, :]
    img = np.real(np.fft.ifft2(np.fft.fft2(f) / (1 + 40 * np.sqrt(fy ** 2 + fx ** 2)) ** 1.5))
    img = img / np.abs(img).max(axis=(1, 2, 3), keepdims=True)
    return torch.from_numpy(((img + 1) * 127.5).clip(0, 255).astype(np.uint8))


def fh_kwargs(data_dir, solver):
    return dict(conditioning_mechanism="online_covariance", cond_scaling=1.0, clip_x0_mean=False,
                max_vector_count=100000, dataset_path=data_dir, image_base_covariance="dct_diagonal",
                denoiser_mean_error_threshold=0.2, use_analytical_score_time_update=True, project_to_diagonal=False,
                space_step_update_threshold=10.0, space_step_update_lower_threshold=1.0, max_rtol=1.0,
                do_space_updates=True, solver_type="customcuda")


def build_net(arch, device, backend, dtype="fp32"):
    from free_hunch_amd import unet as hu
    from free_hunch_amd.precond import iDDPMLinearPrecond
    cfg = {"ffhq": hu.FFHQ256, "imagenet": hu.IMAGENET256}[arch]
    model = hu.UNetModel(cfg, backend=backend, dtype=dtype)
    model.load_state_dict(hu.seeded_state(cfg, 0))
    model = model.to(device).eval()
    return iDDPMLinearPrecond(model, cfg.image_size, 3).to(device), cfg


def run_batch(net, images_u8, seeds, operator_name, num_steps, solver, device, data_dir, groups=1):
    """B independent images through the lock-step sampler; returns uint8 [B,3,S,S] on the device.  The batch runs as
    `groups` lock-step groups on separate host threads / HIP streams, so that the latency-bound Free Hunch phase of one group
    (covariance updates, CG: small kernels with dependent launches) overlaps the MFMA-bound UNet phase of the other.
    Round 2 measured this up to 20x SLOWER: every image then had its own stream for its updates, 8 + 8 streams were multiplexed
    onto 4 hardware queues and the small kernels sat behind convolutions of the other group in the same queue.  Since the
    updates and the CG of a group are single batched launch sequences on the group's own stream (round 3), two groups mean two
    streams: measured 3.43 s vs 3.82 s per batch of 8 (+11 %) on runs of a few batches; four groups of two images lose again
    (UNet at batch 2).  Earlier in round 3 a SUSTAINED run (20 steps) showed nothing (2.200 vs 2.195 images/s), which was read
    as the power limit; it was the device-memory growth of fresh streams per batch, worse with more
    streams.  With the streams kept across batches: 2.27 / 2.31 vs 2.10 / 2.17 images/s over 8 steps, alternating on one box,
    and 2.35 vs 2.20 over 20 steps - two groups are the default (results bitwise equal to one group:
    tests/test_timed_path.py::test_two_lockstep_groups_equal_one_group)."""
    from concurrent.futures import ThreadPoolExecutor
    from free_hunch_amd.measurements import get_operator
    from free_hunch_amd.sampler import StandardRGBEncoder, conditional_sampler_grouped
    enc = StandardRGBEncoder()
    S = images_u8.shape[-1]
    prof = os.environ.get("FH_PHASE_TIMES")
    if prof:
        torch.cuda.synchronize()
    t_setup = time.perf_counter()
    ops, ys, noises = [], [], []
    for b, (img, seed) in enumerate(zip(images_u8, seeds)):
        np.random.seed(int(seed) % (1 << 31))
        torch.manual_seed(int(seed) % (1 << 31))
        op = get_operator(name=operator_name, device=device, sigma_s=0.1, kernel_size=61, intensity=1.0,
                          scale_factor=4, in_shape=(1, 3, S, S),
                          mask_opt={"mask_type": "random", "mask_len_range": (64, 156),
                                    "mask_prob_range": (0.6, 0.8), "image_size": S})
        op.ctx_slot = b
        ops.append(op)
        ys.append(op.forward(enc.encode(img[None].to(device)), noiseless=False))
        noises.append(torch.randn((1, 3, S, S), generator=torch.Generator().manual_seed(int(seed) % (1 << 31)),
                                  dtype=torch.float32))
    B = len(ops)
    if prof:
        torch.cuda.synchronize()
        print(f"[FH_PHASE_TIMES] run_batch setup (operators, measurements, noise): {time.perf_counter() - t_setup:.3f} s", flush=True)
    x = conditional_sampler_grouped(net, torch.cat(noises, 0).to(device), ys, ops, groups=groups, num_steps=num_steps,
                                    sigma_min=0.002, sigma_max=80, rho=7, solver=solver, **fh_kwargs(data_dir, solver))
    run_batch.cg_iters = [sum(t["niter"] for t in m.trace) for m in conditional_sampler_grouped.last_mechanisms]
    return enc.decode(x)


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one (the PMC passes are re-run when a kernel changes)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r00_" + suffix)


def roofline_cov_apply(device, m=32, iters=200, nimg=1):
    """fh_rep_apply at the headline point d = 196608, m = 32 (float64 base), events on the stream.
    nimg > 1: the batched launch the lock-step CG issues (one factor base per image, grid z = image)."""
    import ctypes as C
    from free_hunch_amd import _lib
    S, d = 256, 3 * 256 * 256
    ctx = _lib.Context.get(S, 3 * nimg, 256)
    g = torch.Generator(device="cpu").manual_seed(1)
    Bs = [torch.randn(m, d, generator=g, dtype=torch.float64).to(device) for _ in range(nimg)]
    Ds = [(torch.rand(d, generator=g, dtype=torch.float64) + 0.5).to(device) for _ in range(nimg)]
    rs = [(torch.rand(d, generator=g, dtype=torch.float64) + 0.5).to(device) for _ in range(nimg)]
    Ms = [torch.randn(64, 64, generator=g, dtype=torch.float64).to(device) for _ in range(nimg)]
    z = torch.randn(nimg, d, generator=g, dtype=torch.float64).to(device)
    out = torch.empty_like(z)
    if nimg == 1:
        f = lambda: ctx.rep_apply(Ds[0], rs[0], Bs[0], Ms[0], z[0], out[0], m)
    else:
        per = _lib.FhBatch()
        per.nimg = nimg
        for i in range(nimg):
            per.D[i], per.r[i], per.B[i], per.M[i] = Ds[i].data_ptr(), rs[i].data_ptr(), Bs[i].data_ptr(), Ms[i].data_ptr()
        f = lambda: _lib.check(ctx.lib.fh_rep_apply_batched(ctx.h, C.byref(per), 64, z.data_ptr(), out.data_ptr(), d, m,
                                                            _lib.stream()), "fh_rep_apply_batched")
    def timed():
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 1e3 / iters

    # What the samplers issue: the two-pass kernels (k_rep_dots + k_rep_coef + k_rep_apply2).  The single-sweep kernel
    # (k_rep_fused: B stays in registers between the reduction and the product; opt-in, fh_context_set_exclusive(ctx, 2);
    # bitwise equal, read-once traffic, but not faster: profiles/r02_cov_apply_single_sweep.md) is a profiling variant: it
    # is timed beside the product's kernels only on request (FH_BENCH_SINGLE_SWEEP=1), not in the default bench run.
    ctx.set_exclusive(1)
    sec = timed()
    ctx.status()
    sec_alt = None
    if os.environ.get("FH_BENCH_SINGLE_SWEEP") == "1":
        ctx.set_exclusive(2)
        sec_alt = timed()
        ctx.status()
    ctx.set_exclusive(0)
    algo_bytes = nimg * (8 * d * m + 8 * d * 4)  # per image: base once + D, r, z read + out written (float64)
    achieved = algo_bytes / sec / 1e9
    traffic = None  # HBM-side bytes per apply from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE), if present
    pmc = _latest_profile("cov_apply_pmc.json" if nimg == 1 else "cov_apply_b8_pmc.json")
    if m == 32 and nimg in (1, 8) and os.path.exists(pmc):
        with open(pmc) as f_:
            rec = json.load(f_)
        # the one-image file holds the single-sweep kernel's counters, the two-pass ones under "two_pass_for_comparison"
        traffic = (rec.get("two_pass_for_comparison", {}) if nimg == 1 else rec).get("traffic_bytes_per_apply")
    two_pass, fused = "k_rep_dots + k_rep_coef + k_rep_apply2 (two sweeps of B)", "k_rep_fused<4> (single sweep: B read once)"
    line = {"bound": "hbm", "kernel": f"fh_rep_apply = {two_pass}; d=196608, m={m}, f64, "
                                      f"{nimg} image{'s' if nimg > 1 else ''} per launch",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes": algo_bytes, "us_per_apply": round(sec * 1e6, 2)}
    if sec_alt is not None:
        line["other_variant"] = {"kernel": fused,
                                 "achieved": round(algo_bytes / sec_alt / 1e9, 1), "unit": "GB/s",
                                 "frac": round(algo_bytes / sec_alt / 1e9 / HBM_PEAK_GBS, 4),
                                 "us_per_apply": round(sec_alt * 1e6, 2)}
    return line


def roofline_dense_cov_apply(device, d=12288, iters=30):
    """Dense-matrix covariance path (BASELINE config 3): y = C x with C a float64 (d, d) matrix at the SR
    measurement dimension d = 12288 (1.2 GB) - one streaming pass, 8 d^2 algorithmic bytes - and the rank-2 update
    C <- C + a u v^T + b w z^T (read + write, 16 d^2 bytes).  Events on the launch stream."""
    from free_hunch_amd import dense
    g = torch.Generator().manual_seed(3)
    A = torch.randn(1, d, d, generator=g, dtype=torch.float64).to(device)
    x, u, v = (torch.randn(1, d, generator=g, dtype=torch.float64).to(device) for _ in range(3))
    y = torch.empty_like(x)
    c = torch.full((1,), 1e-3, dtype=torch.float64, device=device)
    out = torch.empty_like(A)

    def timed(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 1e3 / iters

    t_mv = timed(lambda: dense.matvec(A, x, out=y))
    t_r2 = timed(lambda: dense.rank2(A, u, v, c, v, u, c, out=out))
    mv_bytes, r2_bytes = 8 * d * d + 16 * d, 16 * d * d + 32 * d
    ach = mv_bytes / t_mv / 1e9
    traffic = None  # HBM-side bytes per mat-vec from the committed PMC passes (1.0009 x algorithmic)
    pmc = os.path.join(ROOT, "profiles", "r01_dense_pmc.json")
    if d == 12288 and os.path.exists(pmc):
        with open(pmc) as f_:
            traffic = json.load(f_).get("traffic_bytes_per_apply_d12288")
    return {"bound": "hbm", "kernel": f"k_dense_mv (y = C x, d={d}, f64, bs=1)", "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes": mv_bytes, "us_per_apply": round(t_mv * 1e6, 1),
            "rank2_update": {"kernel": "k_dense_rank2", "achieved": round(r2_bytes / t_r2 / 1e9, 1), "unit": "GB/s",
                             "frac": round(r2_bytes / t_r2 / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": r2_bytes,
                             "us_per_update": round(t_r2 * 1e6, 1)}}


def roofline_conv_mfma(device, iters=20):
    """The dominant UNet kernel: 3x3 conv 128 -> 128 on 8 x 256 x 256 NHWC (the most frequent FFHQ layer), events on
    the launch stream.  k_conv_x6r computes the fp32 convolution with six bf16 MFMAs per K = 16 block (exact 3-way
    operand split), so `achieved` counts the EXECUTED bf16 matrix FLOPs (6 x the algorithmic fp32 FLOPs) against the
    dense bf16 peak (2.5 PFLOP/s, MI355X_MICROARCH.md); `fp32_equivalent_tflops` is the algorithmic rate, to be read
    against the 157.3 TFLOP/s fp32 matrix peak that the fp32-MFMA kernel (k_conv_igemm, also timed) is bound by."""
    from free_hunch_amd import _lib
    from free_hunch_amd.unet_hip import _split3
    lib = _lib.load()
    lib.fh_unet_set_precision(0)  # the roofline kernel is the exact-split one, whatever mode the timed run used
    N, H, W, Ci, Co, k = 8, 256, 256, 128, 128, 3
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, H, W, Ci, generator=g).to(device)
    w = (torch.randn(Co, k * k, Ci, generator=g) * 0.03).to(device)
    wx = _split3(w)
    b = torch.zeros(Co, device=device)
    out = torch.empty(N, H, W, Co, device=device)
    f_x6 = lambda: _lib.check(lib.fh_conv2d_x6_nhwc(x.data_ptr(), wx.data_ptr(), b.data_ptr(), None, out.data_ptr(), None,
                                                    1, N, H, W, Ci, Co, k, k, 1, 1, _lib.stream()), "conv x6")
    f_32 = lambda: _lib.check(lib.fh_conv2d_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, 1,
                                                 N, H, W, Ci, Co, k, k, 1, 1, _lib.stream()), "conv f32")

    def timed(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 1e3 / iters

    t6, t32 = timed(f_x6), timed(f_32)
    flops = 2.0 * N * H * W * Ci * Co * k * k
    ach = 6 * flops / t6 / 1e12
    traffic = None  # HBM-side bytes per launch from the committed PMC passes (profiles/tools/pmc_conv.sh), if present
    pmc = _latest_profile("conv_pmc.json")
    if os.path.exists(pmc):
        with open(pmc) as f_:
            per = json.load(f_).get("traffic_bytes_per_launch", {})
        traffic = next((v for k_, v in per.items() if k_.startswith("k_conv_x6r")), None)
    return {"bound": "mfma", "kernel": "k_conv_x6r<256,false,256,3,GL> (weight tile by LDS-DMA) 3x3 128->128 on 8x256x256 NHWC, fp32 via 6 x v_mfma_f32_32x32x16_bf16",
            "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": traffic,
            "algorithmic_bytes": int(2 * N * H * W * 128 * 4 + 3 * 2 * Ci * Co * k * k),
            "flops_per_launch": 6 * flops, "us_per_launch": round(t6 * 1e6, 1),
            "fp32_equivalent_tflops": round(flops / t6 / 1e12, 1),
            "fp32_mfma_kernel": {"kernel": "k_conv_igemm<2,2> (v_mfma_f32_32x32x2_f32)", "achieved": round(flops / t32 / 1e12, 1),
                                 "peak": 157.3, "unit": "TFLOP/s", "frac": round(flops / t32 / 1e12 / 157.3, 4),
                                 "us_per_launch": round(t32 * 1e6, 1)}}


def host_cpu_info():
    """(model name, physical cores, logical CPUs) of the host, from /proc/cpuinfo"""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                logical += 1
            elif ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or logical or (os.cpu_count() or 1)), (logical or (os.cpu_count() or 1))


def cpu_baseline(arch, operator_name, num_steps, data_dir, unet_calls=2, solver="heun"):
    """The oracle (a port of the reference path, pinned to the reference by tests/test_oracle_golden.py) on the host cores,
    on a bounded sample of the bench workload (one full image is ~7 min of CPU time, beyond the bench budget):
      * the Free Hunch side - time / space updates, the CG solve through the operator, the 0.2-std branch - of every guidance
        call of a 10-step trajectory of one 256 x 256 image (same sigma range 80 -> 0.01, space updates included), scaled by
        the call count, with a closed-form Gaussian-prior denoiser standing in for the UNet (the FH side's cost depends on
        sigma and k, not on where the denoiser values come from);
      * the UNet forward + input-VJP of the named architecture, timed on `unet_calls` real calls (its cost does not depend on
        sigma), and one call of the ImageNet-256 architecture for BASELINE configs[0];
    images/s = 1 / (FH side + (2 N - 1) x UNet call)."""
    from oracle import fh_oracle as fo, unet_oracle as uo
    model, phys, logical = host_cpu_info()
    threads = max(1, min(phys, 64))       # UNet convolutions scale to the cores
    fh_threads = max(1, min(phys, 32))    # the FH side (256^2 FFTs / DCTs, [d, k] products) is fastest at <= 32 threads
    torch.set_num_threads(threads)
    S = 256
    kd = os.path.join(ROOT, "free-hunch_amd", "data", "kernels")
    kernel = np.load(os.path.join(kd, "gaussian_ks61_std3.0.npy"))
    op = fo.OracleOperator("gaussian_blur", (1, 3, S, S), 0.1, kernel=kernel)
    img = smooth_images(1, S, 1234)[0]
    x0 = fo.encode_rgb(img[None])
    g = torch.Generator().manual_seed(0)
    y = op.forward(x0, noise=torch.randn(x0.shape, generator=g))
    noise = torch.randn((1, 3, S, S), generator=g, dtype=torch.float32)
    n_calls = 2 * num_steps - 1 if solver == "heun" else num_steps

    def unet_call_seconds(cfg, n):
        net = fo.LinearPrecond(uo.OracleUNet(cfg, uo.seeded_state(cfg, 0)))
        ts = []
        for i in range(n):
            xt = (noise.double() * 3.0).requires_grad_()
            t0 = time.perf_counter()
            d, _ = net(xt, torch.tensor(3.0, dtype=torch.float64))
            torch.autograd.grad((d * x0.double()).sum(), xt)
            ts.append(time.perf_counter() - t0)
        return float(np.mean(ts)), net

    cfg = {"ffhq": uo.FFHQ256, "imagenet": uo.IMAGENET256}[arch]
    t_unet, real = unet_call_seconds(cfg, unet_calls)

    class StandIn:  # posterior mean of a N(0, 0.25 I) prior: linear in x, differentiable, free
        sigma_min, sigma_max, u = real.sigma_min, real.sigma_max, real.u
        round_sigma = staticmethod(real.round_sigma)

        def __call__(self, x, sigma):
            return x * (0.25 / (0.25 + sigma ** 2)), None

    fh_times, iters = [], []

    def fac(op_, v0, d):
        mech = fo.OracleFreeHunch(1.0, op_, False, v0, d, image_base_covariance="dct_diagonal", data_dir=data_dir)

        class Timed:
            def __call__(self, *a):
                t0 = time.perf_counter()
                out = mech(*a)
                fh_times.append(time.perf_counter() - t0)
                iters.append(mech.trace[-1]["niter"])
                return out
        return Timed()

    # bounded sample: a Heun-10 trajectory (19 guidance calls over the same sigma range 80 -> 0.01, space updates included)
    # stands for the 59 calls of Heun-30 - per-call cost is a function of sigma (CG tolerance) and k - and is scaled by the
    # call count; k only reaches 5 instead of 16 here, which favours the CPU slightly
    sample_steps = min(num_steps, 10)
    torch.set_num_threads(fh_threads)
    t0 = time.perf_counter()
    fo.conditional_sampler(StandIn(), noise, y, op, num_steps=sample_steps, solver=solver, mechanism_factory=fac)
    wall_fh = time.perf_counter() - t0
    torch.set_num_threads(threads)
    t_fh = float(np.sum(fh_times)) * n_calls / len(fh_times)
    per_image = t_fh + n_calls * t_unet
    out = {"value": round(1.0 / per_image, 6), "unit": "images/s", "cores": threads, "kind": "port",
           "host_cpu": model, "physical_cores": phys, "logical_cpus": logical,
           "fh_side_threads": fh_threads,
           "sample": f"Free Hunch side of the {len(fh_times)} guidance calls of a {solver}-{sample_steps} trajectory of one "
                     f"256x256 image ({float(np.sum(fh_times)):.1f} s on {fh_threads} threads: sigma 80 -> 0.01, "
                     f"{int(np.sum(iters))} CG iterations; closed-form stand-in denoiser), scaled to {n_calls} calls "
                     f"({t_fh:.0f} s) + {unet_calls} timed {arch.upper()}-256 UNet forward+VJP calls ({t_unet:.2f} s each on "
                     f"{threads} threads) x {n_calls}; {per_image:.0f} s per image, "
                     f"{wall_fh + unet_calls * t_unet:.0f} s of CPU work sampled"}
    if arch != "imagenet":  # BASELINE configs[0]: the ImageNet-256 architecture on the CPU path, one timed UNet call
        t_im, _ = unet_call_seconds(uo.IMAGENET256, 1)
        out["imagenet256_arch"] = {"value": round(1.0 / (t_fh + n_calls * t_im), 6), "unit": "images/s",
                                   "unet_call_s": round(t_im, 2),
                                   "sample": "same Free Hunch side + 1 timed ImageNet-256 UNet forward+VJP call"}
    return out


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def rank_cpu_threads(world):
    """CPU threads per rank: physical cores / world (logical CPUs / 2 where /proc/cpuinfo does not tell), capped at 32."""
    phys = None
    try:
        cores = set()
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    phys = phys or max(1, (os.cpu_count() or 2) // 2)
    return max(1, min(32, phys // max(1, world)))


def launch_ranks(n, argv, script=None, env=None, timeout=None):
    """Start one child process per GPU (the launch model of the reference's torch_utils/distributed.py:19-45: env://
    rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and wait for all of them.  Rank 0's child
    prints the JSON line on the inherited stdout.  Returns the first non-zero child exit code (0 if all succeeded);
    when one rank fails the others are terminated, so a dead rank cannot leave the rest hanging in a collective."""
    import subprocess
    script = os.path.abspath(__file__) if script is None else script
    base = dict(os.environ if env is None else env)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool support only dmabuf IPC; without it RCCL's (and torch's) cross-process
    # buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  Kept if the caller already set it.
    base.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base.get("MASTER_PORT") or _free_port()),
                HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # one rank per GPU shares the host: cap each rank's CPU thread pools at its share of the cores (the host side of a rank
    # is launch-bound Python + a few small torch-CPU ops; 256 OpenMP threads per rank x 8 ranks only fight each other)
    share = str(max(1, rank_cpu_threads(n)))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        base.setdefault(var, share)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e))
    rc, t_end = 0, None if timeout is None else time.monotonic() + timeout
    pending = list(procs)
    while pending:
        for p_ in list(pending):
            code = p_.poll()
            if code is None:
                continue
            pending.remove(p_)
            if code != 0 and rc == 0:
                rc = code
                for q_ in pending:  # exact PIDs we started
                    q_.terminate()
        if pending:
            if t_end is not None and time.monotonic() > t_end:
                rc = rc or 124
                for q_ in pending:
                    q_.kill()
            time.sleep(0.2)
    return rc


def exchange(out_u8, world, coll):
    """The path's one exchange step: ONE all_gather of the finished uint8 images of every rank (inside the timed
    region).  `coll` is the device the collective runs on (the GPU under RCCL; the CPU in the gloo rehearsal)."""
    if world == 1:
        return [out_u8]
    import torch.distributed as dist
    src = out_u8.to(coll)
    bufs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(bufs, src)
    return bufs


def run_steps(step, steps, warmup, world, coll, device_sync):
    """The bench contract's timing: `warmup` untimed steps, then exactly `steps` steps bracketed by device sync + barrier
    on both sides; returns the MAX over ranks of the elapsed seconds."""
    import torch.distributed as dist

    def sync():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    for i in range(warmup):
        step(-1 - i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=coll, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    return elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--groups", type=int, default=2,
                    help="lock-step groups per GPU (default 2: the Free Hunch phase of one group of 4 overlaps the UNet phase of "
                         "the other; 2.35 vs 2.20 images/s over 20 steps - see run_batch; 1 = one group of 8)")
    ap.add_argument("--arch", default="ffhq", choices=["ffhq", "imagenet"])
    ap.add_argument("--operator", default="gaussian_blur")
    ap.add_argument("--num-steps", type=int, default=30)
    ap.add_argument("--solver", default="heun")
    ap.add_argument("--unet-backend", default=os.environ.get("FH_UNET_BACKEND", "hip"))
    ap.add_argument("--unet-dtype", default="fp32", choices=["fp32", "fp16x3", "bf16x3", "bf16", "fp16"],
                    help="bf16: reduced-precision torso (non-parity speed mode, reported separately from the fp32 headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-half-split-leg", action="store_true",
                    help="skip the extra leg that times the same workload with --unet-dtype fp16x3 (N = 1, fp32 runs only)")
    ap.add_argument("--cpu-calls", type=int, default=2, help="real UNet calls timed by the CPU-baseline leg")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher.  Nothing above has touched the GPU (importing torch
        # does not), and the ranks are CHILD processes - a process that has initialised HIP is never re-exec'ed.
        rc = launch_ranks(a.gpus, sys.argv[1:])
        sys.exit(rc if rc >= 0 else 128 - rc)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # FH_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 control flow on a one-GPU box - every rank uses cuda:0 and the
    # collectives go through gloo on host copies (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("FH_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    coll = torch.device("cpu") if rehearsal else device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl")
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()  # no-op when free-hunch_amd/libfh_hip.so is up to date
    if world > 1:
        dist.barrier()  # the other ranks load the library only after rank 0 has (re)built it
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    data_dir = os.path.join(ROOT, "free-hunch_amd", "data")
    net, cfg = build_net(a.arch, device, a.unet_backend, a.unet_dtype)
    images = smooth_images(a.batch, 256, 1234 + rank)

    def step(i):
        seeds = [(i * world + rank) * a.batch + j for j in range(a.batch)]
        out = run_batch(net, images, seeds, a.operator, a.num_steps, a.solver, device, data_dir, a.groups)
        exchange(out, world, coll)
        return out

    elapsed = run_steps(step, a.steps, a.warmup, world, coll, torch.cuda.synchronize)

    if rank == 0:
        total_images = a.steps * a.batch * world
        line = {
            "metric": f"images/sec (256x256, num_steps={a.num_steps}, FH-{a.solver.capitalize()})",
            "value": round(total_images / elapsed, 5), "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp16-compute UNet convolutions (the reference's use_fp16 torso arithmetic; NON-PARITY mode), f32 elsewhere"
                      if a.unet_dtype == "fp16" else
                      "bf16-compute UNet convolutions (NON-PARITY mode), f32 elsewhere" if a.unet_dtype == "bf16" else
                      "UNet convolutions as 3 bf16 products of 2-plane operands (~2^-16, NON-PARITY mode), f32 elsewhere"
                      if a.unet_dtype == "bf16x3" else
                      "f32 UNet with the 3x3 convolutions as 3 half-precision products of 2-plane operands (operands to 1 fp32 ulp, "
                      "dropped term 2^-24 rms: fp32-convolution accuracy against float64, not bit-comparable; opt-in), f32 elsewhere"
                      if a.unet_dtype == "fp16x3" else
                      "f32 UNet (convolutions: exact 3-way bf16 split on the bf16 MFMA, fp32 accuracy)"
                      if os.environ.get("FH_CONV_MODE", "x6") == "x6" else "f32 UNet (fp32 MFMA)") + " + f64 covariance/CG",
            "data": "synthetic",
            "config": {"workload": f"{a.arch.upper()}-256 arch, {a.operator}, FH low-rank covariance (dct_diagonal), "
                                   f"num_steps={a.num_steps} {a.solver}, batch={a.batch} per GPU",
                       "images_per_step": a.batch * world, "lockstep_groups_per_gpu": a.groups,
                       "unet_backend": a.unet_backend,
                       "net_calls_per_image": 2 * a.num_steps - 1 if a.solver == "heun" else a.num_steps,
                       "cg_iters_per_image_last_step": getattr(run_batch, "cg_iters", None)},
        }
        # headline point of SURVEY.md 8(d): d = 196608, m = 32, b = the batch the lock-step CG applies per launch
        line["roofline"] = roofline_cov_apply(device, nimg=min(a.batch, 8), iters=100)
        if a.batch > 1:
            line["roofline_cov_apply_b1"] = roofline_cov_apply(device, nimg=1)
        line["roofline_unet_conv"] = roofline_conv_mfma(device)
        line["roofline_dense_cov_apply"] = roofline_dense_cov_apply(device)
        if world == 1 and a.unet_dtype == "fp32" and a.unet_backend == "hip" and not a.no_half_split_leg:
            # The same workload once more with the opt-in half-split convolutions, measured in the same
            # process on the same box, so that the two figures can be compared without the box-to-box spread.  Reported
            # beside the headline, never as `value`.
            del net
            net2, _ = build_net(a.arch, device, a.unet_backend, "fp16x3")
            k = max(1, min(a.steps, 3))

            def step2(i):
                seeds = [i * a.batch + j for j in range(a.batch)]
                return run_batch(net2, images, seeds, a.operator, a.num_steps, a.solver, device, data_dir, a.groups)

            t2 = run_steps(step2, k, 1, 1, coll, torch.cuda.synchronize)
            line["half_split_mode"] = {
                "value": round(k * a.batch / t2, 5), "unit": "images/s", "steps": k, "warmup": 1,
                "ms_per_step": round(t2 / k * 1e3, 2), "flag": "--unet-dtype fp16x3",
                "dtype": "f32 UNet with the 3x3 convolutions as 3 half-precision products of 2-plane operands, fp32 accumulation "
                         "(fp32-convolution accuracy against float64: tests/test_hip_unet.py::test_half_split_*; not "
                         "bit-comparable with the exact split, hence opt-in) + f64 covariance/CG"}
            del net2
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.arch, a.operator, a.num_steps, data_dir, a.cpu_calls, a.solver)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
