"""Edge cases of the path: the reference's error behaviour at the registries (CPU), C-ABI argument validation, and the
degenerate CG inputs the reference's cg() handles (zero right-hand side, iteration cap, no factor columns), compared
with the oracle's restatement of cg.py:232-282."""
import ctypes as C
import math
import os

import pytest
import torch

import inputs

F64 = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- host registries (no GPU)
def test_registry_errors_match_reference():
    from free_hunch_amd.conditioning_mechanisms import choose_conditioning_mechanism
    from free_hunch_amd.measurements import get_operator
    with pytest.raises(ValueError, match="Unknown conditioning mechanism"):      # conditioning_mechanisms.py:36
        choose_conditioning_mechanism("no_such_mechanism")
    with pytest.raises(ValueError):                                              # DDNM branch :34
        choose_conditioning_mechanism("ddnm")
    assert choose_conditioning_mechanism("tmpd").__name__ == "TMPD"              # per-pixel-variance methods
    assert choose_conditioning_mechanism("peng_convert").__name__ == "PengConvert"
    assert choose_conditioning_mechanism("dps").__name__ == "DPS"                # scalar-variance comparison methods
    assert choose_conditioning_mechanism("pigdm").__name__ == "PiGDM"
    with pytest.raises(NameError, match="is not defined"):                       # measurements.py:37-40
        get_operator("no_such_operator")
    assert choose_conditioning_mechanism("online_covariance").__name__ == "BFGSOnlineUpdate"


def test_rtol_func_matches_reference_schedule():
    """conditioning_mechanisms.py:307-323: 1.0 at sigma = 80, 1e-14 for sigma <= 0.1, no upper clamp (the reference's
    max(min(s, 80), max(0.1, s)) only clamps from below)."""
    from free_hunch_amd.conditioning_mechanisms import rtol_func
    from oracle import fh_oracle as fo
    assert abs(rtol_func(80.0, 1.0) - 1.0) < 1e-12
    assert rtol_func(0.1, 1.0) == pytest.approx(1e-14, rel=1e-9)
    assert rtol_func(0.01, 1.0) == pytest.approx(1e-14, rel=1e-9)
    assert rtol_func(160.0, 1.0) > 1.0  # the ineffective upper clamp is reproduced
    prev = 0.0
    for s in (0.1, 0.2, 0.5, 1.0, 3.0, 10.0, 40.0, 80.0):
        r = rtol_func(s, 1.0)
        assert r == pytest.approx(fo.rtol_func(s, 1.0), rel=1e-12) and r >= prev
        prev = r


# ---------------------------------------------------------------- C ABI argument validation (GPU)
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_c_abi_rejects_bad_arguments(dev):
    """Every entry returns a negative status (FH_EINVAL / FH_ESIZE) instead of launching on inconsistent arguments."""
    from free_hunch_amd import _lib
    lib = _lib.load()
    ctx = _lib.Context.get(64, 3, 32)
    d = 3 * 64 * 64
    v = torch.zeros(d, dtype=F64, device=dev)
    st = _lib.stream()
    assert lib.fh_rep_apply(ctx.h, None, None, None, None, 0, v.data_ptr(), v.data_ptr(), d, 0, st) < 0      # D null
    assert lib.fh_rep_apply(ctx.h, v.data_ptr(), None, None, None, 0, v.data_ptr(), v.data_ptr(), d, 4, st) < 0  # m>0, no B
    assert lib.fh_rep_apply(ctx.h, v.data_ptr(), None, None, None, 0, v.data_ptr(), v.data_ptr(), d + 1, 0, st) < 0  # odd d
    assert lib.fh_dct2d(ctx.h, v.data_ptr(), v.data_ptr(), 4, 0, st) != 0                                     # planes > planes_max
    assert lib.fh_dct2d(ctx.h, None, v.data_ptr(), 3, 0, st) != 0
    x = torch.zeros(1, 8, 8, 48, device=dev)
    assert lib.fh_conv2d_nhwc(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), None, 1, 1, 8, 8, 48, 48, 3, 3, 1, 1, st) < 0
    assert lib.fh_conv2d_x6_nhwc(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), None, 1, 1, 8, 8, 48, 48, 3, 3, 1, 1, st) < 0
    assert lib.fh_conv2d_x6_nhwc(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), None, 4, 1, 8, 8, 64, 64, 3, 3, 1, 1, st) < 0  # split-K without workspace
    assert lib.fh_conv3x3_wino_nhwc(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), 1, 8, 7, 64, 64, st) < 0  # odd W
    A = torch.zeros(1, 4, 4, dtype=F64, device=dev)
    assert lib.fh_dense_matvec(A.data_ptr(), v.data_ptr(), v.data_ptr(), None, 1, 4, 1, 1.0, 0.0, st) < 0   # trans w/o scratch
    assert lib.fh_dense_rank2(A.data_ptr(), A.data_ptr(), 1, 4, v.data_ptr(), None, None, None, None, None, 1.0, 0.0, st) < 0
    per = _lib.FhBatch()
    per.nimg = 0
    assert lib.fh_rep_apply_batched(ctx.h, C.byref(per), 0, v.data_ptr(), v.data_ptr(), d, 0, st) < 0
    torch.cuda.synchronize()


def _cov_and_problem(dev, tmp_path, S, n_steps, name="gaussian_blur"):
    from free_hunch_amd import covariance as hc
    from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2
    from test_hip_parity import _hip_op
    d = 3 * S * S
    # identity prior: A C A^T + s^2 I has condition <= 1 / s^2 = 100, so CG iterates are comparable to 1e-9 (with the
    # DCT prior the system has condition ~ 1e6 and un-converged iterates are rounding-chaotic)
    cov = hc.CovarianceHessianBFGS(1.0, 80.0 ** 2, d, device=dev)
    for what, a in inputs.script(77, (1, 3, S, S), n_steps, 80.0):
        if what == "time":
            cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
        else:
            cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    op = _hip_op(name, S, dev)
    prob, keep = _problem(op, cov, _sigma_y2(op))
    return cov, op, prob, keep


def _amm(ctx, prob):
    from free_hunch_amd import _lib

    def f(u):
        out = torch.empty_like(u)
        _lib.check(ctx.lib.fh_amm(ctx.h, C.byref(prob), u.data_ptr(), out.data_ptr(), _lib.stream()), "amm")
        return out
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("n_steps", [0, 3])
def test_cg_degenerate_inputs_follow_reference_cg(dev, tmp_path, n_steps):
    """cg.py:232-282 semantics on the device: zero right-hand side (pAp <= 1e-16 -> break at k = 1, not optimal),
    iteration cap (niter == maxiter, not optimal), a tolerance met at once, with and without factor columns (m = 0).
    The oracle's cg() drives the SAME device operator (fh_amm), so only the loop logic is compared."""
    from free_hunch_amd import _lib
    from oracle import fh_oracle as fo
    S = 64
    cov, op, prob, keep = _cov_and_problem(dev, tmp_path, S, n_steps)
    assert cov.famC.m == 2 * n_steps
    ctx = cov.ctx
    A = _amm(ctx, prob)
    b = inputs.randn((3 * S * S,), 31).to(dev)
    cases = [("zero_rhs", torch.zeros_like(b), 1e-3, 50), ("cap", b, 1e-12, 3), ("at_once", b, 1e3, 50),
             ("converge", b, 1e-6, 500)]
    for tag, rhs, rtol, maxiter in cases:
        sol = torch.empty_like(rhs)
        info = _lib.FhCgInfo()
        _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), rhs.data_ptr(), sol.data_ptr(), rtol, 0.0, maxiter,
                                       C.byref(info), _lib.stream()), "cg")
        x_ref, i_ref = fo.cg(lambda u: A(u.contiguous()), rhs, rtol=rtol, atol=0.0, maxiter=maxiter)
        assert info.niter == i_ref["niter"], (tag, info.niter, i_ref)
        assert bool(info.optimal) == bool(i_ref["optimal"]), (tag, info.optimal, i_ref)
        scale = max(1.0, float(x_ref.abs().max()))
        assert float((sol - x_ref).abs().max()) < 1e-9 * scale, tag
        if tag == "zero_rhs":
            assert info.niter == 1 and not info.optimal and float(sol.abs().max()) == 0.0
        if tag == "cap":
            assert info.niter == maxiter and not info.optimal
        if tag == "at_once":  # x0 = b is far from the solution on this system; a tolerance above ||r_1|| stops at k = 1
            assert info.niter == 1 and info.optimal


@pytest.mark.gpu
def test_batched_cg_with_a_finished_image(dev, tmp_path):
    """Lock-step batch in which one image has a zero right-hand side (it stops at once) and another a looser system:
    every image gets the iteration count and solution of its own single solve."""
    from free_hunch_amd import _lib
    from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2
    from free_hunch_amd import covariance as hc
    from test_hip_parity import _hip_op
    S, nimg = 64, 3
    d = 3 * S * S
    dv = torch.load(os.path.join(ROOT, "free-hunch_amd", "data", "dct_variance.pt"), weights_only=True)
    torch.save(dv[:, :S, :S].contiguous(), tmp_path / "dct_variance.pt")
    covs, probs, keeps = [], [], []
    for i in range(nimg):
        cov = hc.CovarianceHessianBFGSDCT(str(tmp_path), 80.0 ** 2, d, device=dev, use_precalculated_info=True, ctx_slot=i + 1)
        for what, a in inputs.script(100 + i, (1, 3, S, S), 2, 80.0):
            if what == "time":
                cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
            else:
                cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
        covs.append(cov)
    op = _hip_op("gaussian_blur", S, dev)
    b = torch.stack([inputs.randn((d,), 40).to(dev), torch.zeros(d, dtype=F64, device=dev), inputs.randn((d,), 42).to(dev)])
    singles = []
    for i in range(nimg):
        prob, keep = _problem(op, covs[i], _sigma_y2(op))
        sol = torch.empty(d, dtype=F64, device=dev)
        info = _lib.FhCgInfo()
        ctx = covs[i].ctx
        _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b[i].contiguous().data_ptr(), sol.data_ptr(), 1e-5, 0.0, 400,
                                       C.byref(info), _lib.stream()), "cg")
        singles.append((sol, info.niter, info.optimal))
        keeps.append(keep)
    bctx = _lib.Context.get(S, 3 * nimg, 0, slot=6000)
    shared, keep0 = _problem(op, covs[0], _sigma_y2(op))
    per = _lib.FhBatch()
    per.nimg = nimg
    for i, cov in enumerate(covs):
        per.D[i], per.r[i], per.B[i], per.M[i] = (cov.C.D.data_ptr(), cov.C.r.data_ptr(), cov.famC.B.data_ptr(),
                                                  cov.C.M_dev.data_ptr())
    sol = torch.empty_like(b)
    infos = (_lib.FhCgInfo * nimg)()
    rt = (C.c_double * nimg)(*([1e-5] * nimg))
    fn = bctx.lib.fh_cg_solve_batched
    _lib.check(fn(bctx.h, C.byref(shared), C.byref(per), b.data_ptr(), sol.data_ptr(), rt, 0.0, 400, infos, _lib.stream()),
               "cg batched")
    for i in range(nimg):
        assert infos[i].niter == singles[i][1] and infos[i].optimal == singles[i][2], (i, infos[i].niter, singles[i][1])
        assert float((sol[i] - singles[i][0]).abs().max()) <= 1e-12 * max(1.0, float(singles[i][0].abs().max())), i
    assert infos[1].niter == 1 and not infos[1].optimal


@pytest.mark.gpu
@pytest.mark.parametrize("S,stride,kh,kw,density", [
    (64, 1, 9, 5, 0.5),     # k_conv_tile8, one 64 x 32 tile per row block
    (96, 1, 31, 7, 0.3),    # partial tiles in both directions, tall sparse PSF
    (256, 1, 61, 17, 0.2),  # the motion PSF's extent
    (64, 1, 61, 61, 0.1),   # halo beyond the LDS budget of the 8-output kernel: k_conv_tile
    (64, 2, 7, 7, 1.0),     # k_conv_dec / k_conv_up at every stride the reference's SR kernels cover
    (96, 3, 11, 9, 1.0),
    (64, 4, 25, 25, 1.0),
    (256, 4, 25, 25, 1.0),
    (48, 4, 25, 25, 1.0),   # sizes the tiled SR kernels do not take: k_conv_direct / k_conv_tile
])
def test_conv_circ_matches_a_direct_sum(S, stride, kh, kw, density):
    """fh_conv_circ over every kernel it dispatches to (k_conv_tile8, k_conv_tile, k_conv_dec, k_conv_up, k_conv_direct)
    against the defining circular sums in NumPy, forward and adjoint, random non-symmetric tap sets:
      forward  out[i][j] = sum_t w_t in[(s i - dy_t) mod S][(s j - dx_t) mod S]
      adjoint  out[y][x] = sum_t w_t z[(y + dy_t) mod S][(x + dx_t) mod S],  z = in with s - 1 zeros inserted
    and <A x, y> = <x, A^T y> to rounding."""
    import numpy as np
    from free_hunch_amd import _lib
    from free_hunch_amd.measurements import _TapList
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(S * 131 + stride * 17 + kh)
    k = rng.standard_normal((kh, kw)) * (rng.random((kh, kw)) < density)
    k[0, 0], k[-1, -1] = 0.7, -0.4  # the full extent is always populated
    taps = _TapList(k, dev)
    ctx = _lib.Context.get(S, 3, 0)
    So = S // stride
    x = rng.standard_normal((3, S, S))
    u = rng.standard_normal((3, So, So))
    dy, dx, w = taps.dy.cpu().numpy(), taps.dx.cpu().numpy(), taps.w.cpu().numpy()
    fwd = np.zeros((3, So, So))
    zi = np.zeros((3, S, S))
    zi[:, ::stride, ::stride] = u
    adj = np.zeros((3, S, S))
    for t in range(taps.n):
        fwd += w[t] * np.roll(x, (dy[t], dx[t]), axis=(1, 2))[:, ::stride, ::stride]
        adj += w[t] * np.roll(zi, (-dy[t], -dx[t]), axis=(1, 2))
    xd, ud = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    got_f = ctx.conv(xd, torch.empty(3, So, So, dtype=torch.float64, device=dev), taps, 3, stride, False)
    got_a = ctx.conv(ud, torch.empty(3, S, S, dtype=torch.float64, device=dev), taps, 3, stride, True)
    torch.cuda.synchronize()
    assert np.abs(got_f.cpu().numpy() - fwd).max() < 1e-12 * max(1.0, np.abs(fwd).max())
    assert np.abs(got_a.cpu().numpy() - adj).max() < 1e-12 * max(1.0, np.abs(adj).max())
    lhs, rhs = float((got_f * ud).sum()), float((xd * got_a).sum())
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))
