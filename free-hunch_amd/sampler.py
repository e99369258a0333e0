"""The Free Hunch ODE sampler (reference: generate_conditional.py:38-220): EDM sigma grid snapped to the
network's table, Euler/Heun loop in float64, one plugin instance per image."""
from __future__ import annotations

import numpy as np
import torch

from .conditioning_mechanisms import choose_conditioning_mechanism, solve_customcuda_batched
from .measurements import get_operator


def get_sigma_steps(discretization, num_steps, sigma_min, sigma_max, rho, device=None):
    if discretization != "edm":
        raise NotImplementedError("only the 'edm' discretisation is on the Free Hunch path")
    i = torch.arange(num_steps, dtype=torch.float64, device=device)
    return (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho


class StackedRandomGenerator:
    """One torch.Generator per seed (generate_conditional.py:206-220)."""

    def __init__(self, device, seeds):
        self.generators = [torch.Generator(device).manual_seed(int(seed) % (1 << 32)) for seed in seeds]

    def randn(self, size, **kwargs):
        assert size[0] == len(self.generators)
        return torch.stack([torch.randn(size[1:], generator=gen, **kwargs) for gen in self.generators])

    def randn_like(self, input):
        return self.randn(input.shape, dtype=input.dtype, layout=input.layout, device=input.device)


class StandardRGBEncoder:  # training/encoders.py:61-73
    def encode(self, x):
        return x.to(torch.float32) / 127.5 - 1

    def decode(self, x):
        return (x.to(torch.float32) * 127.5 + 128).clip(0, 255).to(torch.uint8)


def _make_mechanism(o, forward_operator, sigma0, data_dim):
    """The plugin instance of one image, built exactly as generate_conditional.py:120-130 does."""
    return choose_conditioning_mechanism(o["conditioning_mechanism"])(
        o["cond_scaling"], forward_operator, o["clip_x0_mean"], init_denoiser_variance=1,
        init_noise_variance=torch.tensor(sigma0, dtype=torch.float64) ** 2, data_dim=data_dim,
        pigdm_posthoc_scaling=o.get("pigdm_posthoc_scaling", False), max_vector_count=o["max_vector_count"],
        data_dir=o["dataset_path"], image_base_covariance=o["image_base_covariance"],
        pca_component_count=o.get("pca_component_count", 10),
        denoiser_mean_error_threshold=o["denoiser_mean_error_threshold"],
        use_analytical_score_time_update=o["use_analytical_score_time_update"],
        project_to_diagonal=o["project_to_diagonal"], space_step_update_threshold=o["space_step_update_threshold"],
        space_step_update_lower_threshold=o["space_step_update_lower_threshold"], max_rtol=o["max_rtol"],
        do_space_updates=o["do_space_updates"], use_analytic_var_at_end=o.get("use_analytic_var_at_end", False),
        solver_type=o.get("solver_type", "customcuda"), use_rtol_func=o.get("use_rtol_func", False),
        diffpir_lambda=o.get("diffpir_lambda", 10.0))


def conditional_sampler(net, noise, cond_images, operator_kwargs, noise_kwargs=None, labels=None,
                        randn_like=torch.randn_like, num_steps=18, sigma_min=None, sigma_max=None, rho=7,
                        solver="heun", discretization="edm", schedule="linear", scaling="none", S_churn=0, S_min=0,
                        S_max=float("inf"), S_noise=1, measurement=None, operator=None, **other_args):
    """Returns (x_final float64, [x_0], y).  `measurement`/`operator` (not in the reference) inject a fixed y and a
    pre-built operator so that runs are reproducible across devices; otherwise y = A x + noise is drawn here."""
    assert solver in ["euler", "heun"]
    if schedule != "linear" or scaling != "none":
        raise NotImplementedError("only sigma(t)=t, s(t)=1 (the reference's defaults) are on the Free Hunch path")
    forward_operator = operator if operator is not None else get_operator(**operator_kwargs)
    if measurement is None:
        cond_images = forward_operator.forward(cond_images, noiseless=False)
    else:
        cond_images = measurement
    sigma_min = 0.002 if sigma_min is None else sigma_min
    sigma_max = 80 if sigma_max is None else sigma_max
    sigma_min, sigma_max = max(sigma_min, net.sigma_min), min(sigma_max, net.sigma_max)
    sigma_steps = get_sigma_steps(discretization, num_steps, sigma_min, sigma_max, rho, noise.device)
    t_steps = net.round_sigma(sigma_steps)
    t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])
    t_list = [float(t) for t in t_steps]  # host copies: the loop's scalar arithmetic stays in float64 on the CPU

    x_next = noise.to(torch.float64) * t_list[0]
    x_all = [x_next.detach()]
    mech = _make_mechanism(other_args, forward_operator, t_list[0], x_next.shape[1:].numel())
    cov_ctx = getattr(getattr(mech, "covariance_model", None), "ctx", None)
    if cov_ctx is not None:
        # one image, one stream: no other grid-synchronising kernel can share the GPU with this image's Free Hunch work,
        # (include/fh_hip.h: fh_context_set_exclusive; level 1 declares it, level 2 would select the single-sweep apply)
        cov_ctx.set_exclusive(other_args.get("exclusive_device", True))
    y = cond_images.to(noise.device)
    f64 = lambda v: torch.tensor(v, dtype=torch.float64, device=noise.device)
    for i, (t_cur, t_next) in enumerate(zip(t_list[:-1], t_list[1:])):
        x_cur = x_next
        gamma = min(S_churn / num_steps, np.sqrt(2) - 1) if S_min <= t_cur <= S_max else 0
        t_hat = float(net.round_sigma(f64(t_cur + gamma * t_cur)))
        churn = max(t_hat ** 2 - t_cur ** 2, 0.0) ** 0.5
        x_hat = x_cur + churn * S_noise * randn_like(x_cur) if churn > 0 else x_cur
        h = t_next - t_hat
        with torch.enable_grad():
            denoised = mech(x_hat.detach(), net, y, f64(t_hat))
        score = -(x_hat - denoised) / t_hat ** 2
        d_cur = -score * t_hat
        x_prime = x_hat + h * d_cur
        t_prime = t_hat + h  # may differ from t_next by one ulp, exactly as in the reference (:150)
        if solver == "euler" or i == num_steps - 1:
            x_next = x_hat + h * d_cur
        else:
            with torch.enable_grad():
                denoised = mech(x_prime.detach(), net, y, f64(t_prime))
            d_prime = (1 / t_prime) * x_prime - (1 / t_prime) * denoised
            x_next = x_hat + h * (0.5 * d_cur + 0.5 * d_prime)
    conditional_sampler.last_mechanism = mech
    if cov_ctx is not None:
        cov_ctx.status()  # raises if a single-sweep apply gave up waiting for its peers (the result would be invalid)
        cov_ctx.set_exclusive(False)
    return x_next, x_all, cond_images


_IMAGE_STREAMS = {}
_IMAGE_STREAMS_LOCK = __import__("threading").Lock()


def _image_stream(device_index, slot, dev):
    """The HIP stream of lock-step image `slot` on this device (one per slot for the life of the process)."""
    with _IMAGE_STREAMS_LOCK:
        st = _IMAGE_STREAMS.get((device_index, slot))
        if st is None:
            st = _IMAGE_STREAMS[(device_index, slot)] = torch.cuda.Stream(device=dev)
        return st


def conditional_sampler_batched(net, noise, measurements, operators, num_steps=18, sigma_min=None, sigma_max=None,
                                rho=7, solver="heun", slot_base=0, batched_cg=True, **other_args):
    """B independent images advanced in lock-step (BASELINE.json config 2, "batch = 8"): every guidance call runs
    ONE UNet forward and ONE UNet input-VJP over the whole batch, while the per-image Free Hunch work (covariance
    updates, CG solve, branch) runs concurrently on one HIP stream + host thread + scratch context per image.
    Each image keeps its own plugin instance, exactly as in `conditional_sampler`; results are identical to running
    the images one by one up to the batch-size dependence of the UNet's floating-point summation order.

    noise [B,3,S,S] float32; measurements / operators: length-B lists (operator b must carry ctx_slot = slot_base + b;
    several calls may run concurrently from different host threads with disjoint slot ranges)."""
    from concurrent.futures import ThreadPoolExecutor
    assert solver in ["euler", "heun"]
    B = noise.shape[0]
    dev = noise.device
    sigma_min = 0.002 if sigma_min is None else sigma_min
    sigma_max = 80 if sigma_max is None else sigma_max
    sigma_min, sigma_max = max(sigma_min, net.sigma_min), min(sigma_max, net.sigma_max)
    t_steps = net.round_sigma(get_sigma_steps("edm", num_steps, sigma_min, sigma_max, rho, dev))
    t_list = [float(t) for t in t_steps] + [0.0]
    o = other_args
    if o.get("discretization", "edm") != "edm" or o.get("schedule", "linear") != "linear" or o.get("scaling", "none") != "none":
        raise NotImplementedError("only the 'edm' discretisation with sigma(t)=t, s(t)=1 is on the Free Hunch path")
    if o.get("S_churn", 0) != 0:
        raise NotImplementedError("the lock-step sampler is deterministic (S_churn = 0); use conditional_sampler")
    import os as _os
    import time as _time
    _t_mech = _time.perf_counter()
    mechs = []
    for b in range(B):
        assert getattr(operators[b], "ctx_slot", 0) == slot_base + b, "every concurrent image needs its own ctx_slot"
        mechs.append(_make_mechanism(o, operators[b], t_list[0], noise.shape[1:].numel()))
        if hasattr(mechs[-1], "covariance_model"):
            mechs[-1].covariance_model.ctx.set_exclusive(False)  # the per-image streams run concurrently with each other
    # the batched CG runs alone on the GPU unless the caller overlaps several lock-step groups (bench.py --groups > 1)
    exclusive_cg = bool(o.get("exclusive_device", True))
    ys = [m.to(dev) for m in measurements]
    # The per-image streams are created ONCE per (device, slot) and kept: the caching allocator keeps freed blocks per stream,
    # and `torch.cuda.Stream()` hands out the 32 pool streams round-robin - with fresh Stream objects per batch the covariance
    # buffers and UNet activations landed on a different pool stream every batch, no stream ever found its own freed blocks,
    # and the process reserved 10 GB more per batch (0.3-0.5 s of hipMalloc per batch, then a multi-second cache flush at the
    # 288 GB limit after ~25 batches).
    device_key = dev.index if dev.index is not None else torch.cuda.current_device()
    streams = [_image_stream(device_key, slot_base + b, dev) for b in range(B)]
    pool = ThreadPoolExecutor(max_workers=B)
    device_index = dev.index if dev.index is not None else torch.cuda.current_device()

    def fan_out(fn):
        """run fn(b) for every image on its own stream/thread, then make the caller's stream wait for all of them"""
        main = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record(main)

        def job(b):
            torch.cuda.set_device(device_index)
            with torch.cuda.stream(streams[b]):
                streams[b].wait_event(start)
                out = fn(b)
                out.record_stream(main)
                done = torch.cuda.Event()
                done.record(streams[b])
            return out, done

        res = list(pool.map(job, range(B)))
        for _, done in res:
            main.wait_event(done)
        return [r[0] for r in res]

    prof = {"fwd": 0.0, "update": 0.0, "solve": 0.0, "vjp": 0.0, "finish": 0.0} if _os.environ.get("FH_PHASE_TIMES") else None
    if prof is not None:
        prof["mech_host"] = _time.perf_counter() - _t_mech
        torch.cuda.synchronize()
        prof["mechanisms"] = _time.perf_counter() - _t_mech
        prof["mem_reserved_GB"] = round(torch.cuda.memory_reserved() / 2**30, 1)
        prof["mem_alloc_GB"] = round(torch.cuda.memory_allocated() / 2**30, 1)
        prof["n_mallocs"] = torch.cuda.memory_stats().get("num_device_alloc", -1)
        _t_loop = _time.perf_counter()

    # Host <-> device rendezvous after the two UNet passes.  Nothing needs it for correctness (the streams are ordered by
    # events); measured on MI355X it is worth ~5 % of a batch: with the host hundreds of launches ahead, the per-image
    # side streams queue behind the UNet's kernels in the shared hardware queues.  FH_PHASE_SYNC=none disables it.
    sync_after = set(filter(None, _os.environ.get("FH_PHASE_SYNC", "fwd,vjp").split(","))) - {"none"}

    def _tick(name, t0):
        if prof is not None or name in sync_after:
            torch.cuda.current_stream().synchronize()  # (this group's stream only: other groups / ranks keep running)
        if prof is not None:
            prof[name] += _time.perf_counter() - t0
        return _time.perf_counter()

    hist = {"x": None, "m": None}  # the previous call's x_t and denoiser output of the whole batch (contiguous [B,3,S,S])

    def guidance(x, t):
        sigma = torch.tensor(t, dtype=torch.float64, device=dev)
        t0 = _tick("finish", _time.perf_counter()) if (prof is not None or sync_after) else 0.0
        x_t = x.detach().requires_grad_()
        with torch.enable_grad():
            x0_mean, _ = net(x_t, sigma)
        t0 = _tick("fwd", t0)
        x_det, m_det = x_t.detach(), x0_mean.detach()
        if batched_cg and not mechs[0].analytic_now(sigma):
            # covariance updates per image on their own streams, then ONE kernel sequence solves all B systems.
            # `not torch.allclose(x, x_prev)` (:250) of all images with one device -> host transfer
            changed = [None] * B
            prev, prev_m = hist["x"], hist["m"]
            if all(len(mm.xs) != 0 for mm in mechs):
                if prev is None:
                    prev = torch.cat([mm.xs[-1] for mm in mechs], 0)
                close = ((x_det - prev).abs() <= 1e-8 + 1e-5 * prev.abs()).reshape(B, -1).all(dim=1)  # allclose's rule
                changed = [not c for c in close.tolist()]
            # one kernel sequence updates the covariance of all B images (same kernels with the image as a grid dimension);
            # per-image streams only when the batch does not qualify (different column counts, truncation, ...)
            if not (hasattr(mechs[0], "fh_update_batched") and type(mechs[0]).fh_update_batched(
                    mechs, x_det, m_det, sigma, prev, prev_m, changed if changed[0] is not None else None, slot=slot_base)):
                fan_out(lambda b: (mechs[b].fh_update(x_det[b:b + 1], m_det[b:b + 1], sigma, net, x_changed=changed[b]),
                                   x_det[b:b + 1])[1])
            t0 = _tick("update", t0)
            infos = []
            mats = solve_customcuda_batched(operators, ys, [m_det[b:b + 1] for b in range(B)],
                                            [mm.covariance_model for mm in mechs], o["max_rtol"], t, infos,
                                            exclusive=exclusive_cg)
            for b in range(B):
                mechs[b]._rec = dict(infos[b])
        else:
            mats = torch.cat(fan_out(lambda b: mechs[b].fh_solve(x_det[b:b + 1], m_det[b:b + 1], ys[b], sigma, net)), 0)
        hist["x"], hist["m"] = x_det, m_det
        t0 = _tick("solve", t0)
        (g,) = torch.autograd.grad((mats * x0_mean).sum(), x_t)
        t0 = _tick("vjp", t0)
        # the 0.2-std branch statistic (conditioning_mechanisms.py:283) of all images with one device -> host transfer
        stds = (g * sigma.pow(2)).reshape(B, -1).std(dim=1).tolist()
        # images on the "cov" branch get C . mat from ONE batched kernel sequence (batched DCT, apply, IDCT) when the whole
        # batch shares the factor count; the per-image finish is then host bookkeeping + two elementwise ops, no fan-out
        cov_all = None
        is_fh = hasattr(mechs[0], "covariance_model")
        need_cov = [is_fh and mechs[b].fh_branch(None, sigma, stds[b]) == "cov" for b in range(B)]
        if any(need_cov) and len({mm.covariance_model.famC.m for mm in mechs}) == 1:
            from .covariance import CovarianceHessianBFGS
            cov_all = CovarianceHessianBFGS.denoiser_cov_vector_dot_batched([mm.covariance_model for mm in mechs], mats.detach(),
                                                                            slot=slot_base)
        same_branch = is_fh and (all(need_cov) or not any(need_cov)) and (cov_all is not None or not any(need_cov))
        if same_branch and len({mm.cond_scaling for mm in mechs}) == 1 and hasattr(type(mechs[0]), "fh_finish_batched"):
            # the whole batch on one branch (the usual case): three elementwise passes over [B,3,S,S] instead of 5 x B
            out = type(mechs[0]).fh_finish_batched(mechs, g, x_det, m_det, sigma, float(t), "cov" if need_cov[0] else "vjp",
                                                   cov_all)
        elif cov_all is not None or not any(need_cov):
            out = torch.cat([mechs[b].fh_finish(mats[b:b + 1], g[b:b + 1], x_det[b:b + 1], m_det[b:b + 1], sigma,
                                                std=stds[b], cov_mat=cov_all[b:b + 1] if need_cov[b] else None)
                             for b in range(B)], 0)
        else:
            out = torch.cat(fan_out(lambda b: mechs[b].fh_finish(mats[b:b + 1], g[b:b + 1], x_det[b:b + 1], m_det[b:b + 1],
                                                                 sigma, std=stds[b])), 0)
        _tick("finish", t0)
        return out.clip(-1, 1) if o["clip_x0_mean"] else out

    x_next = noise.to(torch.float64) * t_list[0]
    for i, (t_cur, t_next) in enumerate(zip(t_list[:-1], t_list[1:])):
        x_hat, t_hat = x_next, t_cur  # S_churn = 0
        h = t_next - t_hat
        denoised = guidance(x_hat, t_hat)
        d_cur = -(-(x_hat - denoised) / t_hat ** 2) * t_hat
        x_prime = x_hat + h * d_cur
        t_prime = t_hat + h
        if solver == "euler" or i == num_steps - 1:
            x_next = x_hat + h * d_cur
        else:
            denoised = guidance(x_prime, t_prime)
            d_prime = (1 / t_prime) * x_prime - (1 / t_prime) * denoised
            x_next = x_hat + h * (0.5 * d_cur + 0.5 * d_prime)
    pool.shutdown()
    if batched_cg and exclusive_cg:
        from . import _lib
        _lib.Context.get(noise.shape[-1], 3 * B, 0, slot=5000 + 64 * slot_base + B).status()
    if prof is not None:
        torch.cuda.synchronize()
        prof["loop_total"] = _time.perf_counter() - _t_loop
        print("[FH_PHASE_TIMES] seconds per batch:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in prof.items()}, flush=True)
    conditional_sampler_batched.last_mechanisms = mechs
    conditional_sampler_batched.tls.mechanisms = mechs  # per host thread (several groups may run concurrently)
    return x_next


import threading as _threading  # noqa: E402

conditional_sampler_batched.tls = _threading.local()

_GROUP_STREAMS = {}


def conditional_sampler_grouped(net, noise, measurements, operators, groups=2, **kw):
    """`conditional_sampler_batched` over B images as `groups` lock-step groups on separate host threads / HIP streams: the
    latency-bound Free Hunch phase of one group (covariance updates, CG) overlaps the MFMA-bound UNet phase of the other
    (2.35 vs 2.20 images/s at B = 8, two groups, on MI355X; three or four groups lose again: the UNet at batch 2).  Operator b
    must carry ctx_slot = b.  Every group keeps ONE stream for the life of the process (see `_image_stream`).  Returns x
    [B,3,S,S] ordered like the inputs, usable on the caller's current stream; `.last_mechanisms` = the B plugin instances."""
    from concurrent.futures import ThreadPoolExecutor
    B = noise.shape[0]
    groups = max(1, min(int(groups), B))
    dev = noise.device
    dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
    bounds = [round(g * B / groups) for g in range(groups + 1)]
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    if groups > 1:
        kw = dict(kw, exclusive_device=False)  # the groups share the GPU: no grid-synchronising kernel may assume it has it alone

    def run_group(g):
        lo, hi = bounds[g], bounds[g + 1]
        torch.cuda.set_device(dev_index)
        stream = _GROUP_STREAMS.get((dev_index, g))
        if stream is None:
            stream = _GROUP_STREAMS[(dev_index, g)] = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            x = conditional_sampler_batched(net, noise[lo:hi], measurements[lo:hi], operators[lo:hi], slot_base=lo, **kw)
            x.record_stream(main)
            done = torch.cuda.Event()
            done.record(stream)
        return x, done, list(conditional_sampler_batched.tls.mechanisms)

    if groups == 1:  # (one group also runs on its own stream, not the legacy default stream: CG graphs need a capturable one)
        res = [run_group(0)]
    else:
        with ThreadPoolExecutor(max_workers=groups) as pool:
            res = list(pool.map(run_group, range(groups)))
    for _x, done, _m in res:
        main.wait_event(done)
    conditional_sampler_grouped.last_mechanisms = [m for r in res for m in r[2]]
    return torch.cat([r[0] for r in res], 0)
