"""GPU parity at FULL SIZE (256 x 256, d = 196608) against fixtures recorded from the reference itself
(tests/golden/make_golden.py: covariance256 / solver256 / trajectories256 / covariance_trunc), and the free-running
behaviour of the headline configurations at 64 x 64 and 256 x 256.

What "free-running" can and cannot show.  With the DCT prior the blur / inpainting systems have cond ~ 1e6 and are
solved to rtol ~ 1 at high sigma, so the CG iterate at the first threshold crossing, the `> 0.2` std branch and
clamp(+-1) are discontinuities: the reference does not reproduce itself across two CPUs there (measured with the oracle on
the MI355X host against these same recordings, `profiles/r02_free_running_spread.json`).  The discrete sequences that ARE
stable (factor count k per call; the sigma sequence) are asserted exactly; the per-call CG iteration counts and the final
image are compared with bounds taken from that measured CPU-to-CPU spread, and every number is written to
`gpurun_out/free_running_report.json` so that the mismatch rate is reported rather than hidden.
"""
import json
import os

import numpy as np
import pytest
import torch

import inputs
import nets
from test_hip_parity import T, _base_kwargs, _hip_op, maxabs

pytestmark = pytest.mark.gpu
F64 = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "free-hunch_amd", "data")
REPORT = os.path.join(ROOT, "gpurun_out", "free_running_report.json")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _sub_check(got, g, key, sub, tol):
    got = T(got).double()
    ref = T(g[key]).double()
    scale = max(1.0, float(ref.abs().max()))
    assert float((got[..., ::sub, ::sub] - ref).abs().max()) < tol * scale, key
    assert abs(float(got.sum()) - float(g[key + "_sum"])) < tol * max(1.0, float(got.abs().sum())), key
    assert abs(float((got ** 2).sum()) - float(g[key + "_sq"])) < 10 * tol * max(1.0, float(g[key + "_sq"])), key


# ---------------------------------------------------------------- a7-a10 at d = 196608
def test_covariance256_vs_reference_golden(dev, gold):
    """The reference's CovarianceHessianBFGSDCT with the shipped dct_variance.pt through 16 time + 16 space updates
    (k = 16 -> m = 32 columns, the headline point of the cov-apply roofline): predicted mean / score after every time
    update and the cov-apply of a seeded probe after every space update, as strided samples + sum + sum of squares.
    1e-8 relative: float64 on both sides, real-factor vs complex-sqrtm factorisations, 32 chained Woodbury steps."""
    from free_hunch_amd import covariance as hc
    g = gold("covariance256")
    meta = eval(str(g["meta"]))
    shape, sub = meta["shape"], meta["sub"]
    d = int(np.prod(shape[1:]))
    cov = hc.CovarianceHessianBFGSDCT(DATA, meta["sigma0"] ** 2, d, device=dev, use_precalculated_info=True)
    steps = inputs.script(meta["script_seed"], shape, meta["n_steps"], meta["sigma0"], None)
    probe = inputs.randn(shape, meta["probe_seed"]).to(dev)
    for si, (what, a) in enumerate(steps):
        pre = f"{si}_"
        if what == "time":
            mean, score = cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
            _sub_check(mean, g, pre + "mean", sub, 1e-8)
            _sub_check(score, g, pre + "new_score", sub, 1e-8)
        else:
            cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
            _sub_check(cov.denoiser_cov_vector_dot(probe), g, pre + "apply", sub, 1e-8)
        assert cov.k == int(g[pre + "k"])
    assert cov.famC.m == 32



def _ld_solve(A, Bm):
    """longdouble Gaussian elimination with partial pivoting (numpy.linalg has no extended precision): A X = Bm"""
    LD = np.longdouble
    A, X, n = A.astype(LD).copy(), Bm.astype(LD).copy(), A.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]], X[[k, p]] = A[[p, k]], X[[p, k]]
        for i in range(k + 1, n):
            f = A[i, k] / A[k, k]
            A[i, k:] -= f * A[k, k:]
            X[i] -= f * X[k]
    for k in range(n - 1, -1, -1):
        X[k] = (X[k] - A[k, k + 1:] @ X[k + 1:]) / A[k, k]
    return X


def test_forward_time_shift_accuracy_vs_extended_precision(dev, gold, monkeypatch):
    """The time update C <- (C^-1 + s I)^-1 on the states of a real full-size trajectory (gaussian blur, Heun-30, HIP UNet)
    against the same formula in 80-bit arithmetic on the CPU, for the updates below sigma = 0.5.  Consecutive sampler steps
    give nearly dependent factor columns: cond(I + s G M) ~ 1e10 .. 1e13, where a plain float64 elimination - this build
    before the refinement, and the reference's route through numpy / torch inverses - leaves 1e-6 .. 1e-4 in C.  The m x m
    solve of k_woodbury_inner is refined with double-double residuals: < 1e-7 of max|C z| (measured 1e-8)."""
    from free_hunch_amd import covariance as hc
    LD = np.longdouble
    g = gold("trajectories256")
    z = inputs.randn((1, 3, 256, 256), 70, torch.float64).reshape(-1)
    zd, zl = z.to(dev), z.numpy().astype(LD)
    checks = []
    orig = hc.CovarianceHessianBFGSDCT.update_time_step

    def checked(self, x_t, sigma_t, sigma_tnext, score_t, only_covariance=False):
        m = self.famC.m
        pre = None
        pre_h = None
        if m >= 8 and float(sigma_tnext) < 0.5 and len(checks) < 4:
            pre = [t.cpu().numpy() for t in (self.C.D, self.C.r, self.C.M_dev[:m, :m], self.famC.B[:m])]
            mh = self.famH.m
            pre_h = [t.cpu().numpy() for t in (self.H.D, self.H.r, self.H.M_dev[:mh, :mh], self.famH.B[:mh])]
        out = orig(self, x_t, sigma_t, sigma_tnext, score_t, only_covariance)
        if pre is not None:
            sh = float(np.float32(float(sigma_tnext) ** (-2) - float(sigma_t) ** (-2)))
            res = {}
            for T_, solve in ((LD, _ld_solve), (np.float64, np.linalg.solve)):
                D, r, M, B = (a.astype(T_) for a in pre)
                e = 1 / (1 + T_(sh) * D)
                K = np.eye(m, dtype=T_) + T_(sh) * (((B * (r * r * e)) @ B.T) @ M)
                Mp = solve(K.T, M.T).T
                zz = zl.astype(T_)
                res[T_] = D * e * zz + r * e * (B.T @ (Mp @ (B @ (r * e * zz))))
            truth = res[LD]
            got = self._apply(self.C, self.famC, zd, torch.empty_like(zd)).cpu().numpy()
            sc = float(np.abs(truth).max())
            rec = {"sigma_next": float(sigma_tnext), "m": int(m), "cond": float(np.linalg.cond(K.astype(np.float64))),
                   "hip": float(np.abs(got - truth).max()) / sc,
                   "plain_float64": float(np.abs(res[np.float64] - truth).max()) / sc}
            # the Hessian side of the same update (:172-190): score' = H' H^-1 score = (I + s_h H)^-1 score with the OLD H,
            #   (A + W K W^T)^-1 v = A^-1 v - A^-1 W K (I + G K)^-1 W^T A^-1 v,  A = 1 + s_h D_h, W = r_h .* B_h^T, K = s_h M_h,
            # in 80-bit arithmetic, against the predicted score the update returned (transformed to the DCT basis)
            shh = -float(np.float32(float(sigma_tnext) ** 2 - float(sigma_t) ** 2))
            ws = self._fwd(self._vec(score_t)).cpu().numpy()
            got_s = self._fwd(self._vec(out[1])).cpu().numpy()
            resh = {}
            for T_, solve in ((LD, _ld_solve), (np.float64, np.linalg.solve)):
                D, r, M, B = (a.astype(T_) for a in pre_h)
                e = 1 / (1 + T_(shh) * D)
                Kh = T_(shh) * M
                G = (B * (r * r * e)) @ B.T
                v = ws.astype(T_)
                y = solve(np.eye(M.shape[0], dtype=T_) + G @ Kh, (B @ (r * e * v))[:, None])[:, 0]
                resh[T_] = e * v - e * r * (B.T @ (Kh @ y))
            sch = float(np.abs(resh[LD]).max())
            rec.update({"m_h": int(pre_h[2].shape[0]), "hip_score": float(np.abs(got_s - resh[LD]).max()) / sch,
                        "plain_float64_score": float(np.abs(resh[np.float64] - resh[LD]).max()) / sch})
            checks.append(rec)
        return out

    monkeypatch.setattr(hc.CovarianceHessianBFGSDCT, "update_time_step", checked)
    _free_run(g, "gb256_heun30", 256, _small_net(inputs.SMALL_C, int(g["unet_seed"]), dev), dev, DATA, 4, "hip-unet")
    _report("forward_time_shift_vs_longdouble", {"checks": checks})
    assert len(checks) == 4
    assert max(c["hip"] for c in checks) < 1e-7, checks
    # The predicted score (Hessian side, score' = H' (H^-1 score)) against (I + s_h H)^-1 score from the SAME pre-update H in
    # 80-bit arithmetic: measured 1.8e-5 .. 4.5e-5 here, where the same formula in plain float64 leaves 1.7e-7 .. 1.3e-4.  The
    # gap to the covariance side (1e-10 .. 5e-8 above) is the inverse representation: H^-1 is rebuilt from H by every space
    # update, but below sigma = 1 (no more space updates) it only receives diagonal shifts while H takes forward shifts, so
    # H^-1 H drifts from I by ~1e-5 per decade of sigma.  Nothing consumes the prediction there (it feeds a space update at
    # the same noise level only, :262) - reported, and held to 1e-4 so that a regression of the inverse path shows.
    assert max(c["hip_score"] for c in checks) < 1e-4, checks


# ---------------------------------------------------------------- a10: 0 < max_vector_count < k
@pytest.mark.parametrize("tag", ["dct16_max1", "dct16_max3"])
def test_covariance_truncation_vs_reference_golden(dev, gold, tmp_path, tag):
    """`max_vector_count` in {1, 3} (online_update_bfgs.py:233-245, 309-310): the reference keeps the newest columns of its
    sqrtm-mixed factors and re-derives C^-1, H, H^-1; here the same truncation acts on the m x k factor coordinates and the
    kernels see only a new inner matrix.  Outputs of the reference class itself, d = 768."""
    from free_hunch_amd import covariance as hc
    g = gold("covariance_trunc")
    torch.save(T(g["dct_variance16"]), tmp_path / "dct_variance.pt")
    meta = eval(str(g[f"{tag}__meta"]))
    hip = hc.CovarianceHessianBFGSDCT(str(tmp_path), meta["sigma0"] ** 2, 768, device=dev, use_precalculated_info=True,
                                      max_vector_count=meta["kw"]["max_vector_count"])
    steps = inputs.script(meta["script_seed"], meta["shape"], meta["n_steps"], meta["sigma0"], meta["neg"])
    probe = inputs.randn(meta["shape"], meta["probe_seed"]).to(dev)
    ks = []
    for si, (what, a) in enumerate(steps):
        pre = f"{tag}__{si}_"
        if what == "time":
            mean, score = hip.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
            sc = max(1.0, float(np.abs(g[pre + "mean"]).max()))
            assert maxabs(mean, g[pre + "mean"]) < 1e-8 * sc, (si, "mean")
            assert maxabs(score, g[pre + "new_score"]) < 1e-8 * sc, (si, "score")
        else:
            hip.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
        assert hip.k == int(g[pre + "k"])
        ks.append(hip.k)
        ref = g[pre + "apply"]
        assert maxabs(hip.denoiser_cov_vector_dot(probe), ref) < 1e-8 * max(1.0, float(np.abs(ref).max())), (si, "apply")
    assert max(ks) == meta["kw"]["max_vector_count"]  # the cap was reached and held


def test_space_update_accepts_float32_and_cpu_inputs(dev, tmp_path):
    """The reference passes CPU tensors of any float dtype; converted temporaries must stay alive until the launch
    (a freed temporary would be handed to the next conversion and all four pointers would alias)."""
    from free_hunch_amd import covariance as hc
    d = 3 * 64 * 64
    shape = (1, 3, 64, 64)
    a = inputs.script(31, shape, 1, 80.0)[1][1]
    ref = hc.CovarianceHessianBFGS(1, 80.0 ** 2, d, device=dev)
    ref.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    probe = inputs.randn(shape, 32).to(dev)
    want = ref.denoiser_cov_vector_dot(probe)
    assert torch.isfinite(want).all()
    for conv in (lambda t: t.float(), lambda t: t.float().to(dev), lambda t: t):  # f32 CPU, f32 device, f64 CPU
        cov = hc.CovarianceHessianBFGS(1, 80.0 ** 2, d, device=dev)
        cov.update_space_step(conv(a["m0"]), conv(a["m1"]), a["sigma"], conv(a["x"]), conv(a["xn"]))
        got = cov.denoiser_cov_vector_dot(probe)
        assert torch.isfinite(got).all()
        assert maxabs(got, want) < 1e-4 * float(want.abs().max())  # float32 inputs: 1e-7 relative per entry


# ---------------------------------------------------------------- a8: single-sweep cov-apply == two-pass cov-apply
@pytest.mark.parametrize("S,nimg,m", [(256, 1, 32), (256, 8, 32), (256, 8, 2), (256, 3, 17), (256, 2, 40), (256, 8, 64),
                                      (64, 8, 32), (16, 2, 6), (128, 5, 12)])
def test_cov_apply_single_sweep_equals_two_pass_bitwise(dev, S, nimg, m):
    """fh_rep_apply[_batched] on an exclusive context (k_rep_fused: the factor base stays in registers between the
    reduction and the product, one read of B) against the default two-pass kernels: bitwise equal outputs for every column
    count class (m <= 32: 4 columns per wave, m <= 64: 8), ragged column counts, 1..8 images per launch and grids from 1 to
    256 workgroups per image; repeated launches re-arm the counters; no time-out is reported.  Also against float64 torch."""
    import ctypes as C
    from free_hunch_amd import _lib
    d = 3 * S * S
    ctx = _lib.Context.get(S, 3 * nimg, 256, slot=77)
    g = torch.Generator().manual_seed(100 * S + 10 * nimg + m)
    Bs = [torch.randn(m, d, generator=g, dtype=F64).to(dev) for _ in range(nimg)]
    Ds = [(torch.rand(d, generator=g, dtype=F64) + 0.5).to(dev) for _ in range(nimg)]
    rs = [(torch.rand(d, generator=g, dtype=F64) + 0.5).to(dev) for _ in range(nimg)]
    Ms = [torch.randn(64, 64, generator=g, dtype=F64).to(dev) for _ in range(nimg)]
    z = torch.randn(nimg, d, generator=g, dtype=F64).to(dev)
    per = _lib.FhBatch()
    per.nimg = nimg
    for i in range(nimg):
        per.D[i], per.r[i], per.B[i], per.M[i] = Ds[i].data_ptr(), rs[i].data_ptr(), Bs[i].data_ptr(), Ms[i].data_ptr()

    def run():
        out = torch.full_like(z, float("nan"))
        _lib.check(ctx.lib.fh_rep_apply_batched(ctx.h, C.byref(per), 64, z.data_ptr(), out.data_ptr(), d, m, _lib.stream()),
                   "fh_rep_apply_batched")
        torch.cuda.synchronize()
        return out

    ctx.set_exclusive(False)
    two_pass = run()
    ctx.set_exclusive(2)  # 2: the single-sweep kernel also for several images per launch
    try:
        for _ in range(3):  # the last departing workgroup re-arms the counters for the next launch
            fused = run()
            assert torch.equal(fused, two_pass)
        ctx.status()
    finally:
        ctx.set_exclusive(False)
    for i in range(nimg):
        W = (Bs[i] * rs[i][None, :]).T
        ref = Ds[i] * z[i] + W @ (Ms[i][:m, :m] @ (W.T @ z[i]))
        assert float((two_pass[i] - ref).abs().max()) < 1e-10 * float(ref.abs().max())


# ---------------------------------------------------------------- a12: separable blur folded into the DCT passes
@pytest.mark.parametrize("S", [64, 256])
def test_folded_blur_dct_equals_tap_passes(dev, S, monkeypatch):
    """A_mm(u) = sigma_y^2 u + A idct2(C dct2(A^T u)) for the Gaussian blur with the DCT prior: the path that folds the two
    1-D blur passes into the DCT bases (4 dense passes + apply) against the tap-list path (4 blur passes + 4 DCT passes +
    apply).  Same linear map, different rounding: 1e-12 of max|out|.  Also the whole solve: same iteration count and
    solution to 1e-8 at rtol = 1e-8."""
    import ctypes as C
    from free_hunch_amd import _lib, covariance as hc
    from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2, solve_customcuda
    d = 3 * S * S
    dv = torch.load(os.path.join(DATA, "dct_variance.pt"), weights_only=True)[:, :S, :S].contiguous()
    import tempfile
    tmp = tempfile.mkdtemp()
    torch.save(dv, os.path.join(tmp, "dct_variance.pt"))
    cov = hc.CovarianceHessianBFGSDCT(tmp, 80.0 ** 2, d, device=dev, use_precalculated_info=True)
    for what, a in inputs.script(515, (1, 3, S, S), 4, 80.0, sig_end=2.0):
        if what == "time":
            cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
        else:
            cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    op = _hip_op("gaussian_blur", S, dev)
    assert op.folded_bases() is not None
    u = inputs.randn((1, 3, S, S), 516).to(dev).contiguous()
    ctx = cov.ctx
    outs = []
    for no_fold in ("0", "1"):
        monkeypatch.setenv("FH_NO_FOLD", no_fold)
        prob, keep = _problem(op, cov, _sigma_y2(op))
        assert bool(prob.fold_fwd_w) == (no_fold == "0")
        out = torch.empty_like(u)
        _lib.check(ctx.lib.fh_amm(ctx.h, C.byref(prob), u.data_ptr(), out.data_ptr(), _lib.stream()), "amm")
        torch.cuda.synchronize()
        outs.append(out)
    assert maxabs(outs[0], outs[1]) < 1e-12 * float(outs[1].abs().max())
    y = inputs.smooth_image(S, 517).to(dev)
    x0 = (inputs.smooth_image(S, 518) * 0.9).to(F64).to(dev)
    sols, infos = [], []
    for no_fold in ("0", "1"):
        monkeypatch.setenv("FH_NO_FOLD", no_fold)
        info = []
        sols.append(solve_customcuda(op, y, x0, cov, 1.0, 0.3, info, rtol=1e-8))
        infos.append(info[0])
    assert infos[0]["niter"] == infos[1]["niter"] and infos[0]["optimal"]
    assert maxabs(sols[0], sols[1]) < 1e-8 * float(sols[1].abs().max())  # two rtol = 1e-8 solves of the same system


# ---------------------------------------------------------------- a11-a12 at 256 x 256
@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_solver256_vs_reference_golden(dev, gold, name):
    """choose_solver(customcuda) at full size, shipped DCT prior after a scripted 3-pair sequence ending at sigma = 4
    (C up to O(10) against sigma_y^2 = 0.01): a loose solve (sigma_t = 3, rtol 0.12, 13-66 iterations in the reference)
    and a tight one (sigma_t = 0.12, rtol 5.8e-5, 50-210 iterations).  The tight solve pins the VALUE (the converged
    solution does not depend on the path): 2e-4 of max|mat|, i.e. rtol x the residual-to-solution gain of this system.
    Iteration counts within 10 % (they move with summation order at cond ~ 1e3-1e6, also between two CPUs)."""
    from free_hunch_amd import covariance as hc
    from free_hunch_amd.conditioning_mechanisms import choose_solver
    g = gold("solver256")
    meta = eval(str(g["meta"]))
    size, sub = 256, meta["sub"]
    shape, d = (1, 3, size, size), 3 * size * size
    x = inputs.smooth_image(size, meta["image_seed"])
    p = f"{name}_"
    mask = None
    if name == "inpainting":
        bits = np.unpackbits(g[p + "mask"])[: size * size].reshape(1, 1, size, size)
        mask = torch.from_numpy(bits.copy()).float().repeat(1, 3, 1, 1)
    op = _hip_op(name, size, dev, mask)
    y = inputs.solver256_measurement(name, x, mask, meta["noise_seed"]).to(dev)
    cov = hc.CovarianceHessianBFGSDCT(DATA, 80.0 ** 2, d, device=dev, use_precalculated_info=True)
    for what, a in inputs.script(meta["script_seed"], shape, meta["n_pairs"], 80.0, sig_end=meta["sig_end"]):
        if what == "time":
            cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
        else:
            cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    x0_mean = (x + 0.05 * inputs.randn(x.shape, meta["x0_seed"], torch.float32)).to(F64).to(dev)
    for lab in ("hi", "lo"):
        q = f"{p}{lab}_"
        info = []
        mat = choose_solver(name, op, y, x0_mean, None, cov, "customcuda", 1.0, sigma_t=float(g[q + "sigma_t"]),
                            info_out=info)
        ref = T(g[q + "mat_sub"]).double()
        scale = max(1.0, float(ref.abs().max()))
        n_ref = int(g[q + "niter"])
        assert abs(info[0]["niter"] - n_ref) <= 0.1 * n_ref + 1, (q, info[0], n_ref)
        assert info[0]["optimal"] == bool(g[q + "optimal"])
        assert abs(info[0]["rtol"] - float(g[q + "rtol"])) < 1e-12 * float(g[q + "rtol"]) + 1e-300
        err = float((T(mat)[..., ::sub, ::sub].double() - ref).abs().max())
        if lab == "lo":
            assert err < 2e-4 * scale, (q, err, scale)
            assert abs(float((mat.double() ** 2).sum()) - float(g[q + "mat_sq"])) < 1e-3 * float(g[q + "mat_sq"])
        else:  # an un-converged iterate: comparable only when both sides stopped at the same iteration
            assert err < (5e-3 if info[0]["niter"] == n_ref else 2e-1) * scale, (q, err, scale)


# ---------------------------------------------------------------- a1, a6: free-running trajectories
def _small_net(cfg_in, seed, dev):
    from free_hunch_amd import unet as hu
    from free_hunch_amd.precond import iDDPMLinearPrecond
    cfg = hu.UNetConfig(**{k: getattr(cfg_in, k) for k in
                           ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                            "attention_resolutions", "num_heads", "num_head_channels", "use_scale_shift_norm",
                            "resblock_updown", "use_new_attention_order")})
    model = hu.UNetModel(cfg, backend=os.environ.get("FH_UNET_BACKEND", "hip"))
    model.load_state_dict(hu.seeded_state(cfg, seed))
    return iDDPMLinearPrecond(model.to(dev).eval(), cfg.image_size, 3).to(dev)


def _psnr(a, b):
    """PSNR of two [-1, 1] images on the 8-bit scale (peak 2.0)"""
    mse = float(((T(a).double() - T(b).double()) ** 2).mean())
    return 99.0 if mse == 0 else 10 * np.log10(4.0 / mse)


def _report(tag, rec):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        with open(REPORT) as f:
            data = json.load(f)
    data[tag] = rec
    with open(REPORT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def _cpu_oracle_net(cfg, seed, dev):
    """The oracle's CPU UNet behind the product's net interface: the denoiser values then carry the reference's own
    arithmetic (fp32 PyTorch-CPU), so that what differs from the recording is the Free Hunch path alone."""
    from oracle import fh_oracle as fo, unet_oracle as uo
    onet = fo.LinearPrecond(uo.OracleUNet(cfg, uo.seeded_state(cfg, seed)))

    class Net:
        sigma_min, sigma_max, u = onet.sigma_min, onet.sigma_max, onet.u

        def round_sigma(self, s_):
            return onet.round_sigma(torch.as_tensor(s_).cpu()).to(dev)

        def __call__(self, x, s_):
            a, b = onet(x.cpu(), torch.as_tensor(s_).cpu())
            return a.to(dev), b.to(dev)

    return Net()


def _free_run(g, tag, size, net, dev, data_dir, sub, label):
    from free_hunch_amd.sampler import conditional_sampler
    p = tag + "__"
    over = eval(str(g[p + "over"]))
    opname, solver, nsteps = str(g[p + "op"]), str(g[p + "solver"]), int(g[p + "num_steps"])
    _s_img, s_noise = (int(v) for v in g[p + "seeds"])
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
    op = _hip_op(opname, size, dev, mask)
    noise = inputs.randn((1, 3, size, size), s_noise, torch.float32).to(dev)
    y = T(g[p + "y"]).to(dev)
    os.environ["FH_TRACE_SUMS"] = "1"
    try:
        x, _, _ = conditional_sampler(net, noise, None, None, num_steps=nsteps, sigma_min=0.002, sigma_max=80, rho=7,
                                      solver=solver, measurement=y, operator=op, **_base_kwargs(data_dir, over))
    finally:
        os.environ.pop("FH_TRACE_SUMS", None)
    tr = conditional_sampler.last_mechanism.trace
    n_hip, n_ref = np.array([t["niter"] for t in tr]), np.asarray(g[p + "niter"])
    b_hip, b_ref = np.array([int(t["branch"] == "cov") for t in tr]), np.asarray(g[p + "branch_cov"])
    s_hip, s_ref = np.array([t["out_sum"] for t in tr]), np.asarray(g[p + "out_sum"])
    xs = T(x)[..., ::sub, ::sub].double()
    ref = T(g[p + "x_final"]).double()
    # a call "agrees" while the returned estimate's checksum matches the recording to 1e-6 of its size (d^0.5 scale)
    agree = np.abs(s_hip - s_ref) < 1e-6 * max(1.0, (3 * size * size) ** 0.5) + 1e-6 * np.abs(s_ref)
    first_div = int(np.argmin(agree)) if not agree.all() else len(tr)
    rec = {"unet": label, "calls": len(tr), "k_equal": [t["k"] for t in tr] == list(g[p + "k"]),
           "branch_mismatch_calls": int((b_hip != b_ref).sum()), "niter_equal_calls": int((n_hip == n_ref).sum()),
           "first_call_with_a_different_estimate": first_div,
           "niter_sum_hip": int(n_hip.sum()), "niter_sum_ref": int(n_ref.sum()),
           "niter_max_rel_dev": float((np.abs(n_hip - n_ref) / np.maximum(n_ref, 1)).max()),
           "niter_hip": [int(v) for v in n_hip], "niter_ref": [int(v) for v in n_ref],
           "final_max_abs": float((xs - ref).abs().max()), "final_rms": float(((xs - ref) ** 2).mean().sqrt()),
           "final_psnr_vs_ref_db": float(_psnr(xs, ref)), "ref_abs_max": float(ref.abs().max())}
    if p + "x0_sub" in g.files:  # reconstruction quality of both sides against the ground-truth image
        x0 = T(g[p + "x0_sub"]).double()
        rec["psnr_hip_vs_truth_db"], rec["psnr_ref_vs_truth_db"] = float(_psnr(xs, x0)), float(_psnr(ref, x0))
    _report(f"{tag}[{label}]", rec)
    sig = np.array([t["sigma"] for t in tr])
    assert rec["k_equal"], rec
    assert np.allclose(sig, g[p + "sigma"], rtol=2e-7, atol=0)  # the float32 sigma table differs by 1 ulp across hosts
    return rec, tr


def _spread(tag):
    with open(os.path.join(ROOT, "profiles", "r02_free_running_spread.json")) as f:
        return json.load(f)[tag]


# The yardstick for the rounding-chaotic configurations: `profiles/r02_free_running_spread.json` holds the same statistics
# for the ORACLE (the reference's own arithmetic, bit-identical to the reference on the recording host) run on the MI355X
# host's CPU against the recordings - the reference-vs-reference spread between two CPUs.  Measured there: identical k
# sequences, 0-2 calls on the other branch, total CG iterations within 2 %, per-call counts within 16 %, and a final image
# that differs from the recording by 0.4 - 2.0 max-abs (PSNR 20.7 - 40.2 dB; these UNets have random weights, every
# trajectory ends in a saturated +-1 image and a flipped pixel costs 2.0).
FREE_64 = ["gb_heun10", "mb_heun10", "ip_euler20", "gb_heun30"]


@pytest.mark.parametrize("tag", FREE_64)
def test_free_running_headline_configs_64(dev, gold, tag, tmp_path):
    """gaussian_blur / motion_blur / inpainting with the DCT prior (the operators of BASELINE configs[1], [3], [4])
    free-running against the reference's recordings at 64 x 64.  The denoiser values come from the oracle's CPU UNet (the
    reference's arithmetic), so the HIP Free Hunch path is the only difference from the recording - the same experiment as
    the spread file's, with the HIP path in place of the second CPU.  Asserted: exact k and sigma sequences; branch
    decisions, total and per-call CG iterations no further from the recording than 2x beyond what the reference shows
    against itself.  The final image's distance and the device-UNet run are reported next to it."""
    g = gold("trajectories")
    torch.save(T(g["dct_variance64"]), tmp_path / "dct_variance.pt")
    sp = _spread(tag)
    rec, _ = _free_run(g, tag, 64, _cpu_oracle_net(inputs.SMALL_A, int(g["unet_seed"]), dev), dev, tmp_path, 1, "cpu-oracle-unet")
    assert rec["branch_mismatch_calls"] <= 2 * sp["branch_mismatch_calls"] + 2, (rec, sp)
    assert abs(rec["niter_sum_hip"] - rec["niter_sum_ref"]) <= 0.05 * rec["niter_sum_ref"], (rec, sp)
    assert rec["niter_max_rel_dev"] <= 2 * sp["niter_max_rel_dev"] + 0.1, (rec, sp)
    # (the distance of the final image from the recording is REPORTED, not asserted: in these configurations the un-converged
    # high-sigma solves make it a function of the rounding history on both sides - tests/test_cg_sensitivity.py; the value
    # assertions of this path are the call-by-call tests and test_free_running_converged_256_within_1e3)
    rec_dev, _ = _free_run(g, tag, 64, _small_net(inputs.SMALL_A, int(g["unet_seed"]), dev), dev, tmp_path, 1, "hip-unet")
    assert abs(rec_dev["niter_sum_hip"] - rec_dev["niter_sum_ref"]) <= 0.25 * rec_dev["niter_sum_ref"], rec_dev


FREE_256 = [("trajectories256", "gb256_heun30", 16), ("trajectories256", "mb256_heun30", 16),
            ("trajectories256", "sr256_heun30", 16), ("trajectories256", "ip256_heun30", 16),
            ("trajectories256_euler", "ip256_euler100", 28)]


@pytest.mark.parametrize("fixture,tag,k_last", FREE_256)
def test_free_running_trajectory_256_vs_reference_golden(dev, gold, fixture, tag, k_last):
    """one 256 x 256 Heun-30 trajectory per operator, and BASELINE configs[3] (inpainting, Euler,
    num_steps = 100: 100 guidance calls, 28 BFGS pairs -> m = 56 factor columns) - shipped DCT prior, 32-channel UNet with the
    ImageNet-256 block structure (attention at T = 1024 / 256 / 64), recorded from the reference's conditional_sampler, run
    free with the HIP UNet.  The guidance solves of these configurations stop un-converged on cond ~ 1e6 systems
    (tests/test_cg_sensitivity.py: the reference's own iterate moves by 1e-2 under a 1e-16 perturbation) and a random-weight
    UNet amplifies the 1e-5 differences between two fp32 UNet implementations, so the trajectories decorrelate from the
    first call on for every operator (the report lists the first call whose estimate differs: call 0).  Asserted: exact k
    (16 / 28 pairs at the end) and sigma sequences; the first solve's iteration count within 15 % (measured 0 - 12 %); total
    CG work within 5 % and branch decisions within a quarter of the calls; and the same reconstruction statistics as the
    reference against the ground truth (PSNR within 2.5 dB; these are saturated random-UNet outputs at 6 - 8 dB).  Value
    parity of the same path lives in the call-by-call tests (test_teacher_forced_256, test_state_following_euler100_256)
    and, end to end within 1e-3, in test_free_running_converged_256_within_1e3 below."""
    g = gold(fixture)
    rec, tr = _free_run(g, tag, 256, _small_net(inputs.SMALL_C, int(g["unet_seed"]), dev), dev, DATA, 4, "hip-unet")
    assert rec["k_equal"] and tr[-1]["k"] == k_last
    assert abs(rec["niter_hip"][0] - rec["niter_ref"][0]) <= 0.15 * rec["niter_ref"][0] + 1, rec
    assert rec["branch_mismatch_calls"] <= rec["calls"] // 4, rec
    assert abs(rec["niter_sum_hip"] - rec["niter_sum_ref"]) <= 0.05 * rec["niter_sum_ref"], rec
    assert abs(rec["psnr_hip_vs_truth_db"] - rec["psnr_ref_vs_truth_db"]) < 2.5, rec  # measured 0.0 - 1.5 dB


TIGHT_256 = ["gb256_heun12_tight_gauss", "ip256_heun12_tight_damped", "sr256_euler16_tight_damped", "mb256_heun8_tight_gauss"]


@pytest.mark.parametrize("tag", TIGHT_256)
def test_free_running_converged_256_within_1e3(dev, gold, tag):
    """North star: "outputs match the reference PyTorch CPU path on identical seeds within 1e-3 max-abs" - END TO END, free
    running, at full size, against recordings of the reference's `conditional_sampler`, in the configurations where the
    reference reproduces ITSELF across hosts (measured with the oracle under a 1e-16 / 1e-6 perturbation: 5e-7 / 1.5e-4 in
    the final image, against 0.9 - 1.3 with the default max_rtol = 1):
      * max_rtol = 1e-6 (a CLI flag of the reference, config.yaml): every guidance solve converges, so its result is a
        property of the system and no longer of the rounding history;
      * a denoiser that does not amplify: the closed-form Gaussian-prior denoiser (`*_gauss`: exercises the whole Free Hunch
        path, all calls on the vjp branch) or the HIP UNet with a damped output layer (`*_damped`: UNet forward + input-VJP
        through the clamp, both branches).
    Asserted: identical k and branch lists, per-call CG iteration counts within 4 % + 2 (the count of a converged solve at
    cond ~ 1e6 moves with the summation order of its dot products: measured up to 2.5 % over two versions of the p.Ap
    reduction), and the final image within 1e-3 max-abs of the reference's (measured 3e-6 .. 1.4e-4)."""
    g = gold("trajectories256_tight")
    p = tag + "__"
    net = nets.gauss_net(256, dev) if str(g[p + "net"]) == "gauss" else nets.damped_hip_net(inputs.SMALL_C, int(g["unet_seed"]), dev)
    rec, tr = _free_run(g, tag, 256, net, dev, DATA, 2, "hip-" + str(g[p + "net"]))
    assert rec["k_equal"], rec
    assert rec["branch_mismatch_calls"] == 0, rec
    assert all(abs(a - b) <= 0.04 * b + 2 for a, b in zip(rec["niter_hip"], rec["niter_ref"])), rec
    assert rec["final_max_abs"] < 1e-3, rec
    assert abs(float(g[p + "x_final_absmax"]) - rec["ref_abs_max"]) < 1.0  # (the strided sample is representative)


@pytest.mark.parametrize("tag", [t for t in TIGHT_256 if t.endswith("_damped")])
def test_free_running_converged_256_within_1e3_half_split_unet(dev, gold, tag):
    """The same end-to-end bar with the UNet in the opt-in half-split mode (`unet_dtype = fp16x3`: two half-precision planes
    per convolution operand, three products): the two recordings that run the HIP UNet, same assertions - identical k and
    branch lists, iteration counts within 4 % + 2, final image within 1e-3 max-abs of the REFERENCE's recording."""
    g = gold("trajectories256_tight")
    net = nets.damped_hip_net(inputs.SMALL_C, int(g["unet_seed"]), dev, dtype="fp16x3")
    rec, tr = _free_run(g, tag, 256, net, dev, DATA, 2, "hip-damped-fp16x3")
    assert rec["k_equal"], rec
    assert rec["branch_mismatch_calls"] == 0, rec
    assert all(abs(a - b) <= 0.04 * b + 2 for a, b in zip(rec["niter_hip"], rec["niter_ref"])), rec
    assert rec["final_max_abs"] < 1e-3, rec


def test_free_running_sr256_with_reference_unet_arithmetic(dev, gold):
    """The well-conditioned full-size case with the oracle's CPU UNet (the reference's denoiser arithmetic) against the
    reference's 256 x 256 Heun-30 recording (which the oracle reproduces exactly on the recording host:
    tests/test_oracle_golden.py::test_trajectory_256_super_resolution)."""
    g = gold("trajectories256")
    rec, _ = _free_run(g, "sr256_heun30", 256, _cpu_oracle_net(inputs.SMALL_C, int(g["unet_seed"]), dev), dev, DATA, 4,
                       "cpu-oracle-unet")
    sp = _spread("sr256_heun30")
    # 59 calls through a random-weight UNet are not contractive even here: the reference's own arithmetic on this host's CPU
    # (profiles/r02_free_running_spread.json) already differs from the recording in 4 of 59 iteration counts and by 0.82
    # max-abs / 37 dB in the final image.  The HIP path - float64 tap-list operators where the reference blurs through a
    # complex64 OTF - adds perturbations of 1e-7 per solve: it must stay within twice / 20 dB of that spread.
    assert rec["branch_mismatch_calls"] <= 2 * sp["branch_mismatch_calls"] + 3, (rec, sp)
    assert rec["niter_equal_calls"] >= rec["calls"] * 3 // 4, (rec, sp)
    assert abs(rec["niter_sum_hip"] - rec["niter_sum_ref"]) <= 0.02 * rec["niter_sum_ref"], (rec, sp)
    # (final-image distance: reported only, see test_free_running_headline_configs_64)


@pytest.mark.parametrize("opname,tag", [("gaussian_blur", "gb256_heun30"), ("motion_blur", "mb256_heun30"),
                                        ("super_resolution", "sr256_heun30"), ("inpainting", "ip256_heun30")])
def test_teacher_forced_256(dev, gold, opname, tag):
    """Every operator with the shipped DCT prior at full size, call by call: the oracle drives a whole Heun-12 trajectory
    (23 guidance calls, sigma 80 -> 0.01, the inputs of the 256 x 256 fixtures) and the HIP plugin receives the same
    (x_t, denoiser output, y, sigma) at every call while keeping its own covariance state.  Asserted per call:
      * state parity: identical factor count and branch, and the two covariance states agree on a probe vector to 1e-6
        (measured <= 1.5e-7 after 8 space + 11 time updates; below sigma = 0.2 the reference's own arithmetic is 3e-7 ..
        1.4e-6 off the exact update, this build 1e-8: test_forward_time_shift_accuracy_vs_extended_precision);
      * value parity: converged solves (rtol <= 1e-4: the steps that fix the final image) within 1e-5 of max|out|;
      * the un-converged solves (sigma >= 1: the reference stops CG at rtol 0.04 .. 1 on a system of condition ~ 1e6).  Their
        iterate is NOT a property of the system: tests/test_cg_sensitivity.py shows on the reference's own arithmetic that a
        1e-16 perturbation of the right-hand side moves the 40th iterate by 1e-3 .. 1e-1 (and that the complex64 OTF is just
        one such perturbation).  Asserted for every such call instead:
          - the iteration map: both sides stopped after 6 iterations agree to 1e-5 of max|mat| (operators, covariance apply and
            CG recurrences at that state; the reference's complex64 OTF bounds it: measured 2e-8 for the Gaussian PSF, 2e-6 for
            the motion-blur PSF), and to 1e-8 + 200 x the state difference against the oracle with a complex128 OTF (measured:
            <= 2.6e-9 while the factor is empty, up to 60 x the probe difference of the two covariance states afterwards;
            inpainting: no OTF);
          - equal validity: the HIP solution's TRUE residual in the ORACLE's system, ||b - (s^2 I + A C A^T)_oracle u_hip||,
            meets the reference's stopping rule rtol ||b|| as well as the oracle's own iterate does (5 % slack);
          - for the first two such calls both sides are re-solved at rtol 1e-6 and THOSE agree to 1e-5 (measured <= 5e-7);
        the deviation of the un-converged outputs themselves goes to the report next to the oracle's own deviation under the
        1e-16 perturbation (`self_dev`, first three such calls) - same order of magnitude, no assertion."""
    from oracle import fh_oracle as fo, unet_oracle as uo
    from free_hunch_amd.conditioning_mechanisms import BFGSOnlineUpdate, solve_customcuda
    from test_oracle_golden import _mk_op
    g = gold("trajectories256")
    size, ncalls, nsteps = 256, 23, 12
    p = tag + "__"
    s_img, s_noise = (int(v) for v in g[p + "seeds"])
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
    hop, oop = _hip_op(opname, size, dev, mask), _mk_op(opname, size, g, p)
    if opname != "inpainting":
        oop.forward(inputs.smooth_image(size, s_img))
    noise, y = inputs.randn((1, 3, size, size), s_noise, torch.float32), T(g[p + "y"])
    kw = _base_kwargs(DATA, {})
    onet = fo.LinearPrecond(uo.OracleUNet(inputs.SMALL_C, uo.seeded_state(inputs.SMALL_C, int(g["unet_seed"]))))
    probe = inputs.randn((1, 3, size, size), 99, torch.float64)
    rows = []

    class Stop(Exception):
        pass

    def resolve_tight(pair, y_, sigma, rt=1e-6):
        keep, io, ih = fo.rtol_func, [], []
        fo.rtol_func = lambda s_, m_: rt
        try:
            mo = fo.solve_mat(pair.o.op, y_, pair.o.means[-1], pair.o.cov, 1.0, float(sigma), io)
        finally:
            fo.rtol_func = keep
        mh = solve_customcuda(hop, y_.to(dev), pair.h.denoiser_means[-1], pair.h.covariance_model, 1.0, float(sigma), ih,
                              rtol=rt)
        return maxabs(mo, mh) / float(mo.abs().max()), io[0]["niter"], ih[0]["niter"]

    oop128 = None
    if opname != "inpainting":  # the same operator with a complex128 OTF (test-only option of the oracle)
        oop128 = _mk_op(opname, size, g, p)
        oop128.otf_double = True
        oop128.forward(inputs.smooth_image(size, s_img))
    ulp = 1 + 1e-16 * inputs.randn(tuple(y.shape), 5)

    def unconverged(pair, y_, sigma, r):
        """the three checks of an un-converged call (docstring); fills r"""
        s_, cov_o, x0_o = float(sigma), pair.o.cov, pair.o.means[-1]
        x0_h, cov_h = pair.h.denoiser_means[-1], pair.h.covariance_model
        u_h = solve_customcuda.last_solution.clone()  # the solution of THIS call's solve
        m6h = solve_customcuda(hop, y_.to(dev), x0_h, cov_h, 1.0, s_, rtol=1e-300, maxiter=6)
        m6o = fo.solve_mat(pair.o.op, y_, x0_o, cov_o, 1.0, s_, maxiter=6, rtol=0.0)
        r["short"] = maxabs(m6o, m6h) / float(m6o.abs().max())
        m6x = m6o if oop128 is None else fo.solve_mat(oop128, y_, x0_o, cov_o, 1.0, s_, maxiter=6, rtol=0.0)
        r["short_exact_op"] = maxabs(m6x, m6h) / float(m6x.abs().max())
        A_mm, b, _back, _shape = fo.system(pair.o.op, y_, x0_o, cov_o)
        nb = float(b.norm())
        r["res_true_hip"] = float((b - A_mm(T(u_h).double().flatten())).norm()) / nb
        r["res_rec_oracle"] = float(pair.o.trace[-1]["residual_norm"]) / nb
        if sum("self_dev" in q for q in rows) < 3:
            mo = fo.solve_mat(pair.o.op, y_, x0_o, cov_o, 1.0, s_)
            mp = fo.solve_mat(pair.o.op, y_.double() * ulp, x0_o, cov_o, 1.0, s_)
            r["self_dev"] = maxabs(mo, mp) / float(mo.abs().max())

    class Pair:
        def __init__(self, op_, v0, d):
            self.o = fo.OracleFreeHunch(1.0, op_, False, v0, d, image_base_covariance="dct_diagonal", data_dir=DATA)
            self.h = BFGSOnlineUpdate(1.0, hop, False, 1, torch.as_tensor(v0), d, solver_type="customcuda", data_dir=DATA,
                                      **{k: v for k, v in kw.items() if k not in ("conditioning_mechanism", "cond_scaling",
                                                                                  "clip_x0_mean", "dataset_path")})

        def __call__(self, x_t, net, y_, sigma):
            out_o = self.o(x_t, net, y_, sigma)

            def net_dev(x, s_):
                a, b = net(x.cpu(), torch.as_tensor(s_).cpu())
                return a.to(dev), b.to(dev)

            out_h = self.h(x_t.to(dev).clone(), net_dev, y_.to(dev), sigma.to(dev))
            to, th = self.o.trace[-1], self.h.trace[-1]
            co = self.o.cov.denoiser_cov_vector_dot(probe)
            r = dict(sigma=float(sigma), no=to["niter"], nh=th["niter"], bo=to["branch"], bh=th["branch"],
                     ko=to["k"], kh=th["k"], err=maxabs(out_o, out_h), mag=float(out_o.abs().max()), rtol=float(th["rtol"]),
                     cov_probe=maxabs(co, self.h.covariance_model.denoiser_cov_vector_dot(probe.to(dev)))
                     / float(co.abs().max()))
            if r["rtol"] > 2e-2:  # an un-converged solve (sigma >= 1)
                unconverged(self, y_, sigma, r)
                if sum("tight" in q for q in rows) < 2:
                    r["tight"], r["tight_no"], r["tight_nh"] = resolve_tight(self, y_, sigma)
            rows.append(r)
            if len(rows) >= ncalls:
                raise Stop()
            return out_o

    try:
        fo.conditional_sampler(onet, noise, y, oop, num_steps=nsteps, solver="heun",
                               mechanism_factory=lambda op_, v0, d: Pair(op_, v0, d))
    except Stop:
        pass
    assert len(rows) == ncalls
    _report(f"{tag}[teacher-forced Heun-12]", {"calls": ncalls, "k_last": rows[-1]["kh"],
                                             "equal_niter_calls": sum(r["no"] == r["nh"] for r in rows),
                                             "rows": [{k: (round(v, 12) if isinstance(v, float) else v) for k, v in r.items()}
                                                      for r in rows]})
    equal = 0
    for r in rows:
        assert r["ko"] == r["kh"] and r["bo"] == r["bh"], r
        assert r["cov_probe"] < 1e-6, r  # measured <= 1.5e-7; the reference's own float64 arithmetic is 3e-7 .. 1.4e-6 off there
        if "tight" in r:
            assert r["tight"] < 1e-5 and abs(r["tight_no"] - r["tight_nh"]) <= 0.05 * r["tight_no"] + 2, r
        if "short" in r:  # un-converged: iteration map + equal validity (docstring)
            assert r["short"] < 1e-5, r
            assert r["short_exact_op"] < 1e-8 + 200 * r["cov_probe"], r
            assert r["res_true_hip"] <= 1.05 * max(r["rtol"], r["res_rec_oracle"]) + 1e-9, r
        if r["no"] == r["nh"]:
            equal += 1
            if r["rtol"] <= 2e-2:  # sigma <= 0.43: measured <= 2e-6 (rtol 0.011) and <= 3e-7 (rtol <= 4e-4)
                assert r["err"] / r["mag"] < (1e-5 if r["rtol"] <= 1e-3 else 1e-4), r
        else:
            assert abs(r["no"] - r["nh"]) <= 0.1 * r["no"] + 2, r
    assert sum(r["rtol"] <= 1e-4 and r["no"] == r["nh"] for r in rows) >= 4  # the converged tail carries value assertions
    assert equal >= (2 * ncalls) // 3
    assert rows[-1]["kh"] >= 4


def test_state_following_euler100_256(dev, gold, monkeypatch):
    """BASELINE configs[3] on real states: inpainting, Euler, num_steps = 100 at 256 x 256 (100 guidance calls, 28 BFGS pairs
    -> m = 56 factor columns), inputs of the reference's recording (trajectories256_euler.npz).  The HIP sampler runs free
    (HIP UNet, default flags) and the ORACLE's covariance object follows it: every `update_time_step` / `update_space_step`
    the HIP plugin issues is replayed on the oracle with the same tensors.  (Driving the whole trajectory from the oracle, as
    test_teacher_forced_256 does, costs 6 min of CPU for 100 calls; the states this run visits are just as real.)
    Asserted at every update: identical factor counts (k reaches 28, m = 56) and the covariance apply on a probe - the
    quantity every later solve reads - to 1e-6 relative (measured <= 2.9e-7 over all 125 updates; the reference's own float64
    arithmetic is 3e-7 .. 1.4e-6 off the exact update below sigma = 0.2,
    test_forward_time_shift_accuracy_vs_extended_precision).  The predicted mean / score of a time update (the Hessian side,
    :176-190) is CONSUMED only by a space update at the same noise level (:262), i.e. while sigma > 1 (the lower threshold):
    there they agree to 3e-6 (measured <= 9.5e-7 / 1.5e-7 up to k = 28).  Below sigma = 1 both implementations compute the
    prediction and discard it; the two drift apart there as the Hessian's factor columns become dependent (1e-6 at
    sigma = 0.97 ... 1e-3 / 6e-3 at sigma = 0.01, in the report) - dead values, reported, not asserted.  For guidance calls at
    k = 28 whose solve converges (rtol <= 1e-3, every sixth one): the HIP solve against the oracle's `solve_mat` on the same
    state and inputs to 1e-5 of max|mat| with the same iteration count +- 1."""
    from oracle import fh_oracle as fo
    from free_hunch_amd import covariance as hc, conditioning_mechanisms as cm
    from free_hunch_amd.sampler import conditional_sampler, get_sigma_steps
    from test_oracle_golden import _mk_op
    g = gold("trajectories256_euler")
    tag, size = "ip256_euler100", 256
    p = tag + "__"
    d = 3 * size * size
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1)
    hop, oop = _hip_op("inpainting", size, dev, mask), _mk_op("inpainting", size, g, p)
    noise = inputs.randn((1, 3, size, size), int(g[p + "seeds"][1]), torch.float32).to(dev)
    y = T(g[p + "y"])
    net = _small_net(inputs.SMALL_C, int(g["unet_seed"]), dev)
    sigma0 = float(net.round_sigma(get_sigma_steps("edm", 100, 0.002, min(80.0, net.sigma_max), 7, dev))[0])
    ocov = fo.make_covariance("dct_diagonal", DATA, sigma0 ** 2, d)
    probe = inputs.randn((1, 3, size, size), 99, torch.float64)
    rows, solves = [], []
    orig_t, orig_s = hc.CovarianceHessianBFGSDCT.update_time_step, hc.CovarianceHessianBFGSDCT.update_space_step

    def rel(a, b):
        return maxabs(a, b) / max(1e-300, float(T(b).abs().max()))

    def probe_err(self):
        co = ocov.denoiser_cov_vector_dot(probe)
        return rel(self.denoiser_cov_vector_dot(probe.to(dev)), co)

    def time_step(self, x_t, sigma_t, sigma_tnext, score_t, only_covariance=False):
        out = orig_t(self, x_t, sigma_t, sigma_tnext, score_t, only_covariance)
        mo, so = ocov.update_time_step(T(x_t).double(), float(sigma_t), float(sigma_tnext), T(score_t).double(), only_covariance)
        rows.append(dict(what="time", k=self.k, ko=ocov.k, sigma=float(sigma_tnext), mean=rel(out[0], mo), score=rel(out[1], so),
                         probe=probe_err(self) if self.k % 4 == 0 else 0.0))
        return out

    def space_step(self, m0, m1, sigma_t, x, xn):
        orig_s(self, m0, m1, sigma_t, x, xn)
        ocov.update_space_step(T(m0).double(), T(m1).double(), float(sigma_t), T(x).double(), T(xn).double())
        rows.append(dict(what="space", k=self.k, ko=ocov.k, sigma=float(sigma_t), mean=0.0, score=0.0, probe=probe_err(self)))

    monkeypatch.setattr(hc.CovarianceHessianBFGSDCT, "update_time_step", time_step)
    monkeypatch.setattr(hc.CovarianceHessianBFGSDCT, "update_space_step", space_step)
    orig_solve = cm.solve_customcuda

    def solve(operator, y_, x0_mean, cov, max_rtol, sigma_t, info_out=None, **kw_):
        mat = orig_solve(operator, y_, x0_mean, cov, max_rtol, sigma_t, info_out, **kw_)
        if cov.k == 28 and info_out and info_out[-1]["rtol"] <= 1e-3 and len(rows) % 6 == 0:
            io = []
            mo = fo.solve_mat(oop, y, T(x0_mean).double(), ocov, max_rtol, float(sigma_t), io)
            solves.append(dict(sigma=float(sigma_t), nh=info_out[-1]["niter"], no=io[0]["niter"], err=rel(mat, mo)))
        return mat

    monkeypatch.setattr(cm, "solve_customcuda", solve)
    conditional_sampler(net, noise, None, None, num_steps=100, sigma_min=0.002, sigma_max=80, rho=7, solver="euler",
                        measurement=y.to(dev), operator=hop, **_base_kwargs(DATA, {}))
    tr = conditional_sampler.last_mechanism.trace
    _report(f"{tag}[state-following]", {"updates": len(rows), "solves": solves,
                                        "max_mean": max(r["mean"] for r in rows), "max_score": max(r["score"] for r in rows),
                                        "max_probe": max(r["probe"] for r in rows),
                                        "probe_by_k": {str(r["k"]): r["probe"] for r in rows if r["what"] == "space"},
                                        "rows": [{k_: (float("%.3g" % v) if isinstance(v, float) else v) for k_, v in r.items()}
                                                 for r in rows]})
    assert len(tr) == 100 and [t["k"] for t in tr] == list(g[p + "k"]) and tr[-1]["k"] == 28
    assert conditional_sampler.last_mechanism.covariance_model.famC.m == 56
    for r in rows:
        assert r["k"] == r["ko"], r
        assert r["probe"] < 1e-6, r
        if r["sigma"] > 1.0:  # the prediction feeds the space update at this noise level
            # (measured 2.9e-6 .. 3.2e-6 at k = 27 over boxes / runs - the forward-shift drift documented above; 1e-5 is
            # two orders inside the north star's 1e-3 on the prediction)
            assert r["mean"] < 1e-5 and r["score"] < 1e-5, r
    assert len(solves) >= 3, solves
    for q in solves:
        # solves at rtol <= 1e-3 stopped after the same 5 - 9 iterations: the two iterates differ by what the rounding
        # difference of the two implementations is amplified to at that depth (measured 1e-6 .. 1.5e-5; the oracle's own
        # float sums depend on the host's thread count, so the figure moves between boxes)
        assert q["err"] < 1e-4 and abs(q["nh"] - q["no"]) <= 1, q


# ---------------------------------------------------------------- dense path at the configs[2] headline size
def test_dense_path_d12288_vs_oracle(dev):
    """d = 12288 (the SR measurement dimension, 1.2 GB per float64 matrix; BASELINE configs[2]'s dense point): the
    streaming mat-vec and rank-2 kernels against float64 torch on the host, and one `update_bfgs` step (space update:
    C, C^-1, H by the kernels) against the oracle's dense rule.  The two O(d^3) inverses are library calls on both sides
    and are not compared here (d <= 4096 covers them)."""
    from free_hunch_amd import dense
    from oracle import fh_oracle as fo
    d = 12288
    case = inputs.dense_case(91, 1, d, n_steps=1)
    C, Ci = case["C"], case["Ci"]
    x = inputs.randn((1, d), 92)
    Cd = C.to(dev)
    ref = (C @ x[..., None])[..., 0]
    got = dense.matvec(Cd, x.to(dev)).cpu()
    assert float((got - ref).abs().max()) < 1e-12 * d ** 0.5 * max(1.0, float(ref.abs().max()))
    gott = dense.matvec(Cd, x.to(dev), trans=True).cpu()
    assert float((gott - (C.transpose(1, 2) @ x[..., None])[..., 0]).abs().max()) < 1e-12 * d ** 0.5 * max(1.0, float(ref.abs().max()))
    u, v = inputs.randn((1, d), 93), inputs.randn((1, d), 94)
    a = torch.tensor([0.37], dtype=F64)
    r2 = dense.rank2(Cd, u.to(dev), v.to(dev), a.to(dev), v.to(dev), u.to(dev), a.to(dev)).cpu()
    want = C + 0.37 * (u[:, :, None] * v[:, None, :] + v[:, :, None] * u[:, None, :])
    assert float((r2 - want).abs().max()) < 1e-13 * max(1.0, float(want.abs().max()))
    del r2, want
    # one space update through the product's update_bfgs (same argument list as online_update_bfgs.py:414)
    sig = case["sig"]
    mean = case["x"] + sig[0] ** 2 * case["score"]
    dx = 0.3 * case["e1"][0]
    m1 = mean + 0.4 * dx + 0.02 * case["e2"][0]
    n_cov, n_icov, n_hess, _ = dense.update_bfgs(Cd, Ci.to(dev), mean.to(dev), m1.to(dev), lambda t: t, sig[1],
                                                 case["x"].to(dev), dx.to(dev))
    # oracle rule restated without its O(d^3) products: the rank-structured closed forms of the same update
    s = sig[1]
    de = s ** 2 * (m1[0] - mean[0])
    gam = 1 / (dx[0] @ de)
    cdx = C[0] @ dx[0]
    o_cov = C[0] - torch.outer(cdx, cdx) / (dx[0] @ cdx) + torch.outer(de, de) * gam
    assert float((n_cov[0].cpu() - o_cov).abs().max()) < 1e-10 * max(1.0, float(o_cov.abs().max()))
    o_hess = (o_cov / s ** 2 - torch.eye(d, dtype=F64)) / s ** 2
    assert float((n_hess[0].cpu() - o_hess).abs().max()) < 1e-10 * max(1.0, float(o_hess.abs().max()))
    del o_hess
    # (I - g dx de^T) Ci (I - g de dx^T) + g dx dx^T, expanded with two mat-vecs instead of two d^3 products
    a_ = Ci[0] @ de
    b_ = Ci[0].T @ de
    o_icov = (Ci[0] - gam * torch.outer(dx[0], b_) - gam * torch.outer(a_, dx[0])
              + gam ** 2 * (de @ a_) * torch.outer(dx[0], dx[0]) + gam * torch.outer(dx[0], dx[0]))
    assert float((n_icov[0].cpu() - o_icov).abs().max()) < 1e-9 * max(1.0, float(o_icov.abs().max()))
    # the full oracle rule on a 2048-row slice is covered at d = 4096 by test_hip_dense.py; here check the oracle agrees
    # with the expansion on the covariance itself (guards the restatement above)
    o2 = fo.dense_space_update(C[0][:64, :64].clone(), Ci[0][:64, :64].clone(), mean[0][:64], m1[0][:64], s, dx[0][:64])[0]
    cdx64 = C[0][:64, :64] @ dx[0][:64]
    de64 = s ** 2 * (m1[0][:64] - mean[0][:64])
    want64 = C[0][:64, :64] - torch.outer(cdx64, cdx64) / (dx[0][:64] @ cdx64) + torch.outer(de64, de64) / (dx[0][:64] @ de64)
    assert float((o2 - want64).abs().max()) < 1e-10 * max(1.0, float(want64.abs().max()))
