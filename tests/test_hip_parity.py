"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on seeded inputs and against the golden
vectors generated from the reference.  Tolerances are stated per test; discrete outcomes (CG iteration counts,
branch decisions, factor counts) are asserted exactly."""
import os

import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
F64 = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu()
    return torch.from_numpy(np.asarray(a))


def maxabs(a, b):
    return float((T(a).double() - T(b).double()).abs().max())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


# ---------------------------------------------------------------- DCT (a8 wrapper)
@pytest.mark.parametrize("S", [16, 64, 256])
def test_dct_matches_scipy_and_roundtrips(dev, S):
    import scipy.fft
    from free_hunch_amd import _lib
    ctx = _lib.Context.get(S, 3, 256)
    x = inputs.randn((3, S, S), 7 + S)
    X = ctx.dct2d(x.to(dev))
    ref = scipy.fft.dctn(x.numpy(), type=2, norm="ortho", axes=(-2, -1))
    assert maxabs(X, ref) < 1e-12 * max(1.0, np.abs(ref).max())
    back = ctx.dct2d(X, inverse=True)
    assert maxabs(back, x) < 1e-12
    # orthonormality: energy is preserved
    assert abs(float((X ** 2).sum()) - float((x ** 2).sum())) < 1e-9 * float((x ** 2).sum())
    # in-place
    y = x.to(dev).clone()
    ctx.dct2d(y, out=y)
    assert maxabs(y, ref) < 1e-12 * max(1.0, np.abs(ref).max())


# ---------------------------------------------------------------- covariance object (a7-a10)
def _mk_pair(meta, tmp, dev):
    from oracle import fh_oracle as fo
    from free_hunch_amd import covariance as hc
    shape = meta["shape"]
    d = int(np.prod(shape[1:]))
    kw = dict(max_vector_count=meta["kw"].get("max_vector_count"),
              project_to_diagonal=meta["kw"].get("project_to_diagonal", False))
    orc = fo.make_covariance(meta["kind"], tmp, meta["sigma0"] ** 2, d, **kw)
    if meta["kind"] == "identity":
        hip = hc.CovarianceHessianBFGS(1, meta["sigma0"] ** 2, d, device=dev, **kw)
    else:
        hip = hc.CovarianceHessianBFGSDCT(tmp, meta["sigma0"] ** 2, d, device=dev,
                                          use_precalculated_info=(meta["kind"] == "dct_diagonal"), **kw)
    return orc, hip


CASES = [
    dict(kind="identity", shape=(1, 3, 4, 4), kw={}, sigma0=80.0, n=5, neg=2, only_cov=False),
    dict(kind="identity", shape=(1, 3, 4, 4), kw={"project_to_diagonal": True}, sigma0=80.0, n=4, neg=None, only_cov=False),
    dict(kind="identity", shape=(1, 3, 4, 4), kw={"max_vector_count": 0}, sigma0=80.0, n=3, neg=None, only_cov=False),
    dict(kind="dct_diagonal", shape=(1, 3, 16, 16), kw={}, sigma0=80.0, n=6, neg=3, only_cov=False),
    dict(kind="dct_diagonal_noinfo", shape=(1, 3, 16, 16), kw={}, sigma0=80.0, n=4, neg=None, only_cov=False),
    dict(kind="dct_diagonal", shape=(1, 3, 16, 16), kw={}, sigma0=80.0, n=4, neg=None, only_cov=True),
    dict(kind="dct_diagonal", shape=(1, 3, 64, 64), kw={}, sigma0=80.0, n=20, neg=5, only_cov=False),
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_covariance_vs_oracle(dev, gold, tmp_path, ci):
    """Scripted time/space updates: predicted means, scores, cov-applies and (small d) all four dense matrices.
    Tolerance 1e-8 relative: float64 on both sides, different (real vs complex-sqrtm) factorisations."""
    meta = CASES[ci]
    S = meta["shape"][-1]
    dv = T(gold("covariance")["dct_variance16"]) if S == 16 else T(gold("solver")["dct_variance64"])
    torch.save(dv, tmp_path / "dct_variance.pt")
    orc, hip = _mk_pair(meta, str(tmp_path), dev)
    hip.ctx.set_exclusive(2 * (ci % 2))  # every other case on the single-sweep apply
    try:
        steps = inputs.script(900 + ci, meta["shape"], meta["n"], meta["sigma0"], meta["neg"])
        probe = inputs.randn(meta["shape"], 950 + ci)
        tol = 1e-8 if meta["n"] <= 6 else 1e-7  # 20 chained updates: conditioning of the inner matrices accumulates
        for si, (what, a) in enumerate(steps):
            if what == "time":
                mo, so = orc.update_time_step(a["x"], a["sigma"], a["sigma_next"], a["score"], only_covariance=meta["only_cov"])
                mh, sh = hip.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev),
                                              only_covariance=meta["only_cov"])
                sc = max(1.0, float(mo.abs().max()))
                assert maxabs(mh, mo) < tol * sc, (si, "mean")
                assert maxabs(sh, so) < tol * sc, (si, "score")
            else:
                if meta["only_cov"]:
                    continue
                orc.update_space_step(a["m0"], a["m1"], a["sigma"], a["x"], a["xn"])
                hip.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
            assert hip.k == orc.k
            if not meta["only_cov"] and not meta["kw"]:
                # the inverse representations are state too (read by the step-wise / m > 64 / truncating paths): C (C^-1 z) = z and
                # H (H^-1 z) = z in the transform domain, for an exclusive context as well (single-sweep apply inside the update)
                zz = probe.to(dev).double().reshape(-1)
                for rep_a, rep_b, fam in ((hip.C, hip.Ci, hip.famC), (hip.H, hip.Hi, hip.famH)):
                    t_ = hip._apply(rep_b, fam, zz, torch.empty_like(zz))
                    back = hip._apply(rep_a, fam, t_, torch.empty_like(zz))
                    assert maxabs(back, zz) < 1e-7 * float(zz.abs().max()), (si, "inverse consistency", fam is hip.famC)
            ro = orc.denoiser_cov_vector_dot(probe)
            rh = hip.denoiser_cov_vector_dot(probe.to(dev))
            assert maxabs(rh, ro) < tol * max(1.0, float(ro.abs().max())), (si, "apply")
            if S == 4:
                for nm, a_, b_ in zip("C Ci H Hi".split(), hip.get_dense_matrices(), orc.dense()):
                    b_ = b_.real
                    if not (meta["kind"] == "identity"):
                        continue
                    assert maxabs(a_, b_) < 1e-7 * max(1.0, float(b_.abs().max())), (si, nm)
    finally:
        hip.ctx.set_exclusive(0)


def test_covariance_switches_between_fused_and_stepwise_updates(dev, gold, tmp_path, monkeypatch):
    """The one-call updates (fh_cov_time_update / fh_cov_space_update: forward time shift, closed-form C^-1) and the
    step-by-step path (Woodbury through the inverse representation; taken beyond 64 columns) share the four representations:
    a state built by one must be a valid input of the other.  First 10 scripted steps fused, the rest step-wise, vs the oracle."""
    meta = CASES[6]
    torch.save(T(gold("solver")["dct_variance64"]), tmp_path / "dct_variance.pt")
    orc, hip = _mk_pair(meta, str(tmp_path), dev)
    steps = inputs.script(906, meta["shape"], meta["n"], meta["sigma0"], meta["neg"])
    probe = inputs.randn(meta["shape"], 956)
    for si, (what, a) in enumerate(steps):
        if si == 10:
            monkeypatch.setenv("FH_COV_STEPWISE", "1")
        if what == "time":
            mo, so = orc.update_time_step(a["x"], a["sigma"], a["sigma_next"], a["score"])
            mh, sh = hip.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
            sc = max(1.0, float(mo.abs().max()))
            assert maxabs(mh, mo) < 1e-7 * sc and maxabs(sh, so) < 1e-7 * sc, si
        else:
            orc.update_space_step(a["m0"], a["m1"], a["sigma"], a["x"], a["xn"])
            hip.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
        ro = orc.denoiser_cov_vector_dot(probe)
        assert maxabs(hip.denoiser_cov_vector_dot(probe.to(dev)), ro) < 1e-7 * max(1.0, float(ro.abs().max())), si
    assert hip.k == orc.k and hip.k >= 8



def test_covariance_vs_reference_golden(dev, gold, tmp_path):
    """The dct16 case of covariance.npz (outputs of the reference class itself)."""
    from free_hunch_amd import covariance as hc
    g = gold("covariance")
    torch.save(T(g["dct_variance16"]), tmp_path / "dct_variance.pt")
    for tag in ("dct16", "dct16_noinfo", "dct16_onlycov"):
        meta = eval(str(g[f"{tag}__meta"]))
        hip = hc.CovarianceHessianBFGSDCT(str(tmp_path), meta["sigma0"] ** 2, 768, device=dev,
                                          use_precalculated_info=(meta["kind"] == "dct_diagonal"))
        steps = inputs.script(meta["script_seed"], meta["shape"], meta["n_steps"], meta["sigma0"], meta["neg"])
        probe = inputs.randn(meta["shape"], meta["probe_seed"]).to(dev)
        for si, (what, a) in enumerate(steps):
            pre = f"{tag}__{si}_"
            if what == "time":
                mean, score = hip.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev),
                                                   only_covariance=meta["only_cov"])
                sc = max(1.0, float(np.abs(g[pre + "mean"]).max()))
                assert maxabs(mean, g[pre + "mean"]) < 1e-8 * sc
                assert maxabs(score, g[pre + "new_score"]) < 1e-8 * sc
            else:
                if meta["only_cov"]:
                    continue
                hip.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
            assert hip.k == int(g[pre + "k"])
            ref = g[pre + "apply"]
            assert maxabs(hip.denoiser_cov_vector_dot(probe), ref) < 1e-8 * max(1.0, float(np.abs(ref).max()))


# ---------------------------------------------------------------- operators (a13)
def _hip_op(name, size, dev, mask=None):
    from free_hunch_amd.measurements import get_operator
    kw = dict(name=name, device=dev, sigma_s=0.1, kernel_size=61, intensity=1.0, scale_factor=4,
              in_shape=(1, 3, size, size),
              mask_opt={"mask_type": "random", "mask_len_range": (64, 156), "mask_prob_range": (0.6, 0.8),
                        "image_size": size})
    if mask is not None:
        kw["mask"] = mask
    return get_operator(**kw)


@pytest.mark.parametrize("size", [64, 256])
@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_operators_vs_reference_golden(dev, gold, name, size):
    g = gold("operators")
    p = f"{name}_{size}_"
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if name == "inpainting" else None
    op = _hip_op(name, size, dev, mask)
    x = inputs.smooth_image(size, 5).to(dev)
    y = op.forward(x.clone(), noiseless=True)
    yt = inputs.randn(y.shape, 77, torch.float32).to(dev)
    xt = op.transpose(yt.clone())
    st = size // 32
    ysub = y if (size == 64 or name == "super_resolution") else y[..., ::st, ::st]
    xsub = xt if size == 64 else xt[..., ::st, ::st]
    assert y.dtype == torch.float32 and xt.dtype == torch.float32
    assert maxabs(ysub, g[p + "y"]) < 2e-6          # float32 outputs, c64 FFT in the reference
    assert maxabs(xsub, g[p + "xt"]) < 2e-5
    assert abs(float((y.double() ** 2).sum()) - float(g[p + "y_sq"])) < 1e-3 * max(1.0, float(g[p + "y_sq"]))
    assert abs(float((xt.double() ** 2).sum()) - float(g[p + "xt_sq"])) < 1e-4 * max(1.0, float(g[p + "xt_sq"]))


@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution"])
def test_operator_adjoint_identity_f64(dev, name):
    """<A x, y> = <x, A^T y> in float64 at 256x256 (size-independent property)."""
    op = _hip_op(name, 256, dev)
    x = inputs.randn((1, 3, 256, 256), 1).to(dev)
    stride = 4 if name == "super_resolution" else 1
    ax = op._conv(x, stride=stride)
    y = inputs.randn(tuple(ax.shape), 2).to(dev)
    aty = op._conv(y, stride=stride, adjoint=True)
    lhs, rhs = float((ax * y).sum()), float((x * aty).sum())
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))


# ---------------------------------------------------------------- solver calls (a11-a12)
@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_solver_vs_reference_golden(dev, gold, name, tmp_path):
    from free_hunch_amd import covariance as hc
    from free_hunch_amd.conditioning_mechanisms import choose_solver
    g = gold("solver")
    size = 64
    torch.save(T(g["dct_variance64"]), tmp_path / "dct_variance.pt")
    p = f"{name}_"
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if name == "inpainting" else None
    op = _hip_op(name, size, dev, mask)
    x = inputs.smooth_image(size, 9)
    y = T(g[p + "y"]).to(dev)
    cov = hc.CovarianceHessianBFGSDCT(str(tmp_path), 80.0 ** 2, 3 * size * size, device=dev,
                                      use_precalculated_info=True)
    steps = inputs.script(int(g["script_seed"]), (1, 3, size, size), 3, 80.0)
    for si, (what, a) in enumerate(steps):
        if what == "time":
            cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
        else:
            cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
        x0_mean = (x + 0.05 * inputs.randn(x.shape, 600 + si, torch.float32)).to(F64).to(dev)
        for lab in ("hi", "lo"):
            q = f"{p}{si}_{lab}_"
            info = []
            mat = choose_solver(name, op, y, x0_mean, None, cov, "customcuda", 1.0, sigma_t=float(g[q + "sigma_t"]),
                                info_out=info)
            ref = T(g[q + "mat_sub"])
            if lab == "lo":
                # rtol = 1e-14 is below what float64 CG can reach on this system: both sides iterate until
                # pAp <= 1e-16 / stagnation, the count is rounding noise (it differs between two CPUs running
                # the reference itself); the converged solution is what is comparable
                assert abs(info[0]["niter"] - int(g[q + "niter"])) <= 0.1 * int(g[q + "niter"]) + 5, (q, info[0])
                assert maxabs(mat[..., ::2, ::2], ref) < 1e-6 * max(1.0, float(ref.abs().max())), q
                continue
            # iteration counts at loose tolerances on this cond ~ 1e6 system move with summation order (see the
            # trajectory tests); they must stay within 10 %
            assert abs(info[0]["niter"] - int(g[q + "niter"])) <= 0.1 * int(g[q + "niter"]) + 1, (q, info[0])
            assert info[0]["optimal"] == bool(g[q + "optimal"])
            # the reference blurs through a complex64 OTF (6e-8 relative per frequency); an un-converged CG iterate
            # (rtol up to 1) amplifies that by cond(A C A^T + s^2 I) ~ 1e5, hence 5e-3 rather than 1e-6
            same = info[0]["niter"] == int(g[q + "niter"])
            assert maxabs(mat[..., ::2, ::2], ref) < (5e-3 if same else 1e-1) * max(1.0, float(ref.abs().max())), q


def test_cg_full_size_residual_property(dev, tmp_path):
    """At 256x256 with m = 32 columns: the returned solution satisfies the stopping rule it reports
    (||b - A_mm x|| <= rtol ||b||), checked with an independent fh_amm application."""
    import ctypes as C
    from free_hunch_amd import _lib, covariance as hc
    from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2
    S, d = 256, 3 * 256 * 256
    dv = torch.load(os.path.join(ROOT, "free-hunch_amd", "data", "dct_variance.pt"), weights_only=True)
    torch.save(dv, tmp_path / "dct_variance.pt")
    cov = hc.CovarianceHessianBFGSDCT(str(tmp_path), 80.0 ** 2, d, device=dev, use_precalculated_info=True)
    steps = inputs.script(4242, (1, 3, S, S), 16, 80.0)
    for what, a in steps:
        if what == "time":
            cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
        else:
            cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    assert cov.famC.m == 32
    op = _hip_op("gaussian_blur", S, dev)
    prob, keep = _problem(op, cov, _sigma_y2(op))
    b = inputs.randn((1, 3, S, S), 5).to(dev).contiguous()
    sol = torch.empty_like(b)
    info = _lib.FhCgInfo()
    ctx = cov.ctx
    rtol = 1e-6
    _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b.data_ptr(), sol.data_ptr(), rtol, 0.0, 5000, C.byref(info),
                                   _lib.stream()), "cg")
    assert info.optimal == 1 and 1 <= info.niter < 5000
    ax = torch.empty_like(b)
    _lib.check(ctx.lib.fh_amm(ctx.h, C.byref(prob), sol.data_ptr(), ax.data_ptr(), _lib.stream()), "amm")
    res = float((b - ax).norm())
    assert res <= 1.05 * rtol * float(b.norm())
    assert abs(res - info.residual_norm) < 1e-3 * rtol * float(b.norm())
    # symmetry of A_mm (needed by CG): <u, A v> = <A u, v>
    u = inputs.randn((1, 3, S, S), 6).to(dev).contiguous()
    au = torch.empty_like(u)
    _lib.check(ctx.lib.fh_amm(ctx.h, C.byref(prob), u.data_ptr(), au.data_ptr(), _lib.stream()), "amm")
    l, r = float((u * ax).sum()), float((au * sol).sum())
    assert abs(l - r) < 1e-9 * max(abs(l), 1.0)


# ---------------------------------------------------------------- whole trajectories (a1, a6)
def _hip_net(gold_traj, dev, backend):
    from free_hunch_amd import unet as hu
    from free_hunch_amd.precond import iDDPMLinearPrecond
    cfg = hu.UNetConfig(**{k: getattr(inputs.SMALL_A, k) for k in
                           ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                            "attention_resolutions", "num_heads", "num_head_channels", "use_scale_shift_norm",
                            "resblock_updown", "use_new_attention_order")})
    model = hu.UNetModel(cfg, backend=backend)
    model.load_state_dict(hu.seeded_state(cfg, int(gold_traj["unet_seed"])))
    model = model.to(dev).eval()
    return iDDPMLinearPrecond(model, cfg.image_size, 3).to(dev)


def _base_kwargs(tmp, over):
    base = dict(conditioning_mechanism="online_covariance", cond_scaling=1.0, clip_x0_mean=False,
                max_vector_count=100000, dataset_path=str(tmp), image_base_covariance="dct_diagonal",
                denoiser_mean_error_threshold=0.2, use_analytical_score_time_update=True, project_to_diagonal=False,
                space_step_update_threshold=10.0, space_step_update_lower_threshold=1.0, max_rtol=1.0,
                do_space_updates=True)
    return {**base, **over}


def _cpu_oracle_net(dev):
    """The oracle's CPU UNet behind the product's net interface: bit-identical denoiser values for both sides."""
    from oracle import fh_oracle as fo, unet_oracle as uo
    onet = fo.LinearPrecond(uo.OracleUNet(inputs.SMALL_A, uo.seeded_state(inputs.SMALL_A, 11)))

    class Net:
        sigma_min, sigma_max, u = onet.sigma_min, onet.sigma_max, onet.u

        def round_sigma(self, s):
            return onet.round_sigma(torch.as_tensor(s).cpu()).to(dev)

        def __call__(self, x, s):
            a, b = onet(x.cpu(), torch.as_tensor(s).cpu())
            return a.to(dev), b.to(dev)

    return Net(), onet


# Free-running parity against the reference's own recorded trajectories is meaningful where the path is
# well-conditioned: SR x4 (n = d/16) and the identity prior.  With the DCT prior (variances 6e-4 .. 7e3 against
# sigma_y^2 = 1e-2) the blur/inpainting systems have cond ~ 1e6 and are solved to rtol ~ 1 at high sigma: the CG
# iterate at the first threshold crossing is rounding-chaotic, and the reference does not reproduce ITSELF across
# two CPUs there (tests/test_oracle_golden.py run on the MI355X host: different iteration counts, O(1) final
# differences).  Those configurations are pinned call by call instead (teacher-forced test below).
@pytest.mark.parametrize("tag", ["sr_heun10", "gb_heun10_identity"])
@pytest.mark.parametrize("unet", ["cpu-oracle", "device"])
def test_trajectory_free_running_vs_reference_golden(dev, gold, tag, unet, tmp_path):
    from free_hunch_amd.sampler import conditional_sampler
    g = gold("trajectories")
    p = tag + "__"
    torch.save(T(g["dct_variance64"]), tmp_path / "dct_variance.pt")
    over = eval(str(g[p + "over"]))
    opname, solver, nsteps = str(g[p + "op"]), str(g[p + "solver"]), int(g[p + "num_steps"])
    _s_img, s_noise = (int(v) for v in g[p + "seeds"])
    net = _cpu_oracle_net(dev)[0] if unet == "cpu-oracle" else _hip_net(g, dev, os.environ.get("FH_UNET_BACKEND", "hip"))
    op = _hip_op(opname, 64, dev)
    noise = inputs.randn((1, 3, 64, 64), s_noise, torch.float32).to(dev)
    y = T(g[p + "y"]).to(dev)
    x, _, _ = conditional_sampler(net, noise, None, None, num_steps=nsteps, sigma_min=0.002, sigma_max=80, rho=7,
                                  solver=solver, measurement=y, operator=op, **_base_kwargs(tmp_path, over))
    tr = conditional_sampler.last_mechanism.trace
    assert [t["k"] for t in tr] == list(g[p + "k"])
    assert [int(t["branch"] == "cov") for t in tr] == list(g[p + "branch_cov"])
    assert np.allclose([t["sigma"] for t in tr], g[p + "sigma"], rtol=2e-7, atol=0)  # table differs by 1 f32 ulp across hosts
    assert [t["niter"] for t in tr] == list(g[p + "niter"])
    assert maxabs(x, g[p + "x_final"]) < 1e-3  # north-star tolerance



# every recorded configuration runs by default (the CPU-oracle side of the eleven cases costs ~4 min on the box's host)
TF_TAGS = ["gb_heun10", "sr_heun10", "ip_euler20", "gb_heun10_identity", "sr_heun10+analytic", "sr_heun10+netscore"]
if True:
    TF_TAGS += ["mb_heun10", "sr_heun10+project", "gb_heun10_nospace", "gb_heun10_readme", "ip_euler20+analytic"]
# Heun-72 on the SR inputs: 37 BFGS pairs -> 74 factor columns.  Beyond 64 columns the covariance updates leave the one-call
# kernels for the step-by-step path (Woodbury through the inverse representation, m x m solve on the host with
# extended-precision refinement) and the CG applies run on the wide-column kernels
TF_TAGS += ["sr_heun10+long72"]


@pytest.mark.parametrize("tag", TF_TAGS)
def test_trajectory_teacher_forced_vs_oracle(dev, gold, tag, tmp_path):
    """The oracle drives the sampler; at every guidance call the HIP plugin receives the same (x_t, denoiser output,
    y, sigma) and keeps its own covariance state.  Per call: identical factor count and vjp/cov branch; for
    sigma <= 3 (the steps that determine the final image) identical CG iteration counts and outputs within 1e-5 of
    max|out| (measured: 1e-6 .. 1e-10); above that, within 1e-4 whenever both solves are short (<= 20 iterations,
    sigma < 20), i.e. outside the rounding-chaotic regime described above.  The long un-converged solves (their iterate
    moves by 1e-2 under a 1e-16 perturbation in the reference's own arithmetic, tests/test_cg_sensitivity.py) are held to
    what IS reproducible: the iterates after 6 iterations agree to 1e-5 of max|mat| (iteration map at that state), and the
    HIP solution's true residual in the ORACLE's system meets the reference's stopping rule (5 % slack)."""
    from oracle import fh_oracle as fo
    from free_hunch_amd.conditioning_mechanisms import BFGSOnlineUpdate, solve_customcuda
    from test_oracle_golden import _mk_op
    g = gold("trajectories")
    analytic = tag.endswith("+analytic")  # use_analytic_var_at_end = true (scalar-variance closed form below sigma 0.2)
    netscore = tag.endswith("+netscore")  # use_analytical_score_time_update = false (extra UNet call at x_prev, :252-254)
    project = tag.endswith("+project")    # project_to_diagonal = true (:274-277, incl. the Hessian quirk)
    long72 = tag.endswith("+long72")
    tag = tag.split("+")[0]
    p = tag + "__"
    torch.save(T(g["dct_variance64"]), tmp_path / "dct_variance.pt")
    over = eval(str(g[p + "over"]))
    if analytic:
        over = {**over, "use_analytic_var_at_end": True}
    if netscore:
        over = {**over, "use_analytical_score_time_update": False}
    if project:
        over = {**over, "project_to_diagonal": True}
    recon = torch.load(os.path.join(ROOT, "free-hunch_amd", "data", "recon_mse.pt"), weights_only=True)
    opname = str(g[p + "op"])
    s_img, s_noise = (int(v) for v in g[p + "seeds"])
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
    hop, oop = _hip_op(opname, 64, dev, mask), _mk_op(opname, 64, g, p)
    if opname != "inpainting":
        oop.forward(inputs.smooth_image(64, s_img))
    noise, y = inputs.randn((1, 3, 64, 64), s_noise, torch.float32), T(g[p + "y"])
    kw = _base_kwargs(tmp_path, over)
    _, onet = _cpu_oracle_net(dev)
    rows = []
    probe = inputs.randn((1, 3, 64, 64), 97, torch.float64)

    class Pair:
        def __init__(self, op_, v0, d):
            self.o = fo.OracleFreeHunch(1.0, op_, False, v0, d, image_base_covariance=kw["image_base_covariance"],
                                        data_dir=str(tmp_path), do_space_updates=kw["do_space_updates"],
                                        space_step_update_threshold=kw["space_step_update_threshold"],
                                        space_step_update_lower_threshold=kw["space_step_update_lower_threshold"],
                                        use_analytic_var_at_end=analytic, recon_mse=recon,
                                        use_analytical_score_time_update=not netscore, project_to_diagonal=project)
            self.h = BFGSOnlineUpdate(1.0, hop, False, 1, torch.as_tensor(v0), d, solver_type="customcuda",
                                      data_dir=str(tmp_path), **{k: v for k, v in kw.items() if k not in
                                                                 ("conditioning_mechanism", "cond_scaling",
                                                                  "clip_x0_mean", "dataset_path")})

        def __call__(self, x_t, net, y_, sigma):
            out_o = self.o(x_t, net, y_, sigma)

            def net_dev(x, s):
                a, b = net(x.cpu(), torch.as_tensor(s).cpu())
                return a.to(dev), b.to(dev)

            out_h = self.h(x_t.to(dev).clone(), net_dev, y_.to(dev), sigma.to(dev))
            to, th = self.o.trace[-1], self.h.trace[-1]
            rows.append(dict(sigma=float(sigma), no=to["niter"], nh=th["niter"], bo=to["branch"], bh=th["branch"],
                             ko=to["k"], kh=th["k"], err=maxabs(out_o, out_h), mag=float(out_o.abs().max()),
                             res=th["residual_norm"], rtol=th["rtol"], opt=th["optimal"]))
            if long72:
                co = self.o.cov.denoiser_cov_vector_dot(probe)
                rows[-1]["probe"] = maxabs(co, self.h.covariance_model.denoiser_cov_vector_dot(probe.to(dev))) / float(co.abs().max())
            elif th["rtol"] > 2e-2 and to["niter"] > 20 and not th.get("analytic"):
                s_, r = float(sigma), rows[-1]
                u_h = solve_customcuda.last_solution.clone()
                m6h = solve_customcuda(hop, y_.to(dev), self.h.denoiser_means[-1], self.h.covariance_model, 1.0, s_, rtol=1e-300,
                                       maxiter=6)
                m6o = fo.solve_mat(self.o.op, y_, self.o.means[-1], self.o.cov, 1.0, s_, maxiter=6, rtol=0.0)
                r["short"] = maxabs(m6o, m6h) / float(m6o.abs().max())
                A_mm, b, _back, _shape = fo.system(self.o.op, y_, self.o.means[-1], self.o.cov)
                r["res_true_hip"] = float((b - A_mm(T(u_h).double().flatten())).norm()) / float(b.norm())
                r["res_rec_oracle"] = float(to["residual_norm"]) / float(b.norm())
            return out_o

    fo.conditional_sampler(onet, noise, y, oop, num_steps=72 if long72 else int(g[p + "num_steps"]), solver=str(g[p + "solver"]),
                           mechanism_factory=lambda op_, v0, d: Pair(op_, v0, d))
    assert len(rows) == (143 if long72 else len(g[p + "niter"]))
    if long72:
        assert max(r["kh"] for r in rows) > 32, "the run must cross 64 factor columns"
    tight = loose = 0
    if long72 and os.environ.get("FH_TEST_DUMP"):
        for i, r in enumerate(rows):
            print("ROW", i, {k_: (float(f"{v:.3g}") if isinstance(v, float) else v) for k_, v in r.items()}, flush=True)
    for r in rows:
        assert r["ko"] == r["kh"], r
        assert r["bo"] == r["bh"], r
        rel = r["err"] / r["mag"]
        if analytic and r["sigma"] < 0.2:
            assert rel < 1e-5, r  # closed form vs CG on the same system (iteration counts are not comparable)
            tight += 1
        elif long72:
            # With a random-weight UNet and 72 Heun steps the reference's covariance becomes INDEFINITE from the third pair on
            # (its own CG stops after one iteration with p.Ap <= 1e-16 and returns estimates of magnitude 1e2 .. 1e3: rows with
            # no = 1, optimal False); the next space updates divide by q = dx.C dx near its zero crossing and turn the 1e-8
            # difference of the two states into 1e-2.  Value parity is therefore asserted while the state is sane (k <= 10:
            # probe <= 2e-8 measured); after that the run must stay finite, keep the reference's factor count and branch at
            # every call and cross 64 columns - the functional check of the wide path.
            assert np.isfinite(r["err"]) and np.isfinite(r["mag"]), r
            if r["kh"] <= 10:
                assert r["probe"] < 1e-6, r
                if r["no"] == r["nh"] and r["no"] > 1:
                    assert rel < 1e-4, r
                    tight += 1
            else:
                loose += 1
        elif r["sigma"] <= 3.0:
            assert r["no"] == r["nh"] and rel < 1e-5, r
            tight += 1
        elif r["no"] == r["nh"] and r["no"] <= 20 and r["sigma"] < 20:
            assert rel < 1e-4, r
            tight += 1
        elif "short" in r:
            # an un-converged iterate of a long solve at high sigma (rtol 0.04 .. 1): see the docstring
            assert r["short"] < 1e-5, r
            assert r["res_true_hip"] <= 1.05 * max(r["rtol"], r["res_rec_oracle"]) + 1e-9, r
            loose += 1
    # every call with equal iteration counts carries a value assertion; at least half of all calls must be of that kind
    assert tight + loose >= len(rows) // 2, (tight, loose, len(rows))
    assert tight >= (len(rows) // 3 if not long72 else 40)


def test_dct_variance_prior_matches_scipy(dev):
    """Row f1 of SURVEY section 8: the producer of dct_variance.pt on the HIP DCT, against SciPy on the host."""
    import scipy.fft
    from free_hunch_amd.frequency_analysis import dct_variance
    g = torch.Generator().manual_seed(4)
    imgs = torch.randint(0, 256, (20, 3, 64, 64), generator=g, dtype=torch.uint8)
    got = dct_variance(imgs, device=dev, batch=8).cpu().numpy()
    x = imgs.numpy().astype(np.float64) / 127.5 - 1
    ref = (scipy.fft.dctn(x, type=2, norm="ortho", axes=(-2, -1)) ** 2).mean(0)
    assert np.abs(got - ref).max() < 1e-5 * ref.max()


@pytest.mark.parametrize("name", ["gaussian_blur", "super_resolution", "inpainting"])
def test_analytic_var_solver_vs_reference_closed_form(dev, gold, name):
    """`use_analytic_var_at_end`: the reference's Fourier closed forms (conditioning_mechanisms.py:357, :454, :608,
    restated in the oracle) against the HIP solve of the same system, C = theta I, rtol 1e-10."""
    from oracle import fh_oracle as fo
    from free_hunch_amd.covariance import ScalarCovariance
    from free_hunch_amd.conditioning_mechanisms import solve_customcuda
    from test_oracle_golden import _mk_op
    g = gold("solver")
    p = f"{name}_"
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if name == "inpainting" else None
    hop, oop = _hip_op(name, 64, dev, mask), _mk_op(name, 64, g, p)
    x = inputs.smooth_image(64, 9)
    oop.forward(x.clone())
    y = T(g[p + "y"])
    x0_mean = (x + 0.05 * inputs.randn(x.shape, 601, torch.float32)).to(F64)
    theta = 0.0123
    ref = fo.analytic_mat(oop, y, x0_mean, theta)
    info = []
    mat = solve_customcuda(hop, y.to(dev), x0_mean.to(dev), ScalarCovariance(theta, 3 * 64 * 64, dev), 1.0, 0.1, info,
                           rtol=1e-10)
    # the closed form uses the complex64 OTF; agreement to 1e-5 of max|mat| (float32-level operator rounding)
    assert maxabs(mat, ref) < 1e-5 * float(ref.abs().max())


def test_rep_apply_batched_matches_single(dev):
    """fh_rep_apply_batched (grid z = image, pass 2 in reverse image order) is bit-identical to per-image fh_rep_apply."""
    import ctypes as C
    from free_hunch_amd import _lib
    S, nimg, m = 64, 5, 12
    d = 3 * S * S
    ctx = _lib.Context.get(S, 3 * nimg, 64)
    g = inputs.rng(91)
    Bs = [torch.randn(m, d, generator=g, dtype=F64).to(dev) for _ in range(nimg)]
    Ds = [(torch.rand(d, generator=g, dtype=F64) + 0.5).to(dev) for _ in range(nimg)]
    rs = [(torch.rand(d, generator=g, dtype=F64) + 0.5).to(dev) for _ in range(nimg)]
    Ms = [torch.randn(16, 16, generator=g, dtype=F64).to(dev) for _ in range(nimg)]
    z = torch.randn(nimg, d, generator=g, dtype=F64).to(dev)
    per = _lib.FhBatch()
    per.nimg = nimg
    for i in range(nimg):
        per.D[i], per.r[i], per.B[i], per.M[i] = Ds[i].data_ptr(), rs[i].data_ptr(), Bs[i].data_ptr(), Ms[i].data_ptr()
    out = torch.empty_like(z)
    _lib.check(ctx.lib.fh_rep_apply_batched(ctx.h, C.byref(per), 16, z.data_ptr(), out.data_ptr(), d, m, _lib.stream()),
               "fh_rep_apply_batched")
    for i in range(nimg):
        one = torch.empty(d, dtype=F64, device=dev)
        ctx.rep_apply(Ds[i], rs[i], Bs[i], Ms[i], z[i].contiguous(), one, m)
        assert torch.equal(one, out[i]), i
        ref = Ds[i] * z[i] + rs[i] * (Bs[i].T @ (Ms[i][:m, :m] @ (Bs[i] @ (rs[i] * z[i]))))
        assert maxabs(one, ref) < 1e-9 * float(ref.abs().max())


@pytest.mark.parametrize("m", [1, 2, 7, 32, 64])
def test_woodbury_inner_on_device_matches_numpy(dev, m):
    """fh_woodbury_inner (Gauss-Jordan with partial pivoting in one workgroup) against the host formula
    -M (I + G M)^-1 (numpy LU), on Gram-like G and an indefinite block-diagonal-plus-noise M as the updates produce."""
    from free_hunch_amd import _lib
    from free_hunch_amd.covariance import _woodbury_inner
    ctx = _lib.Context.get(16, 3, 64)
    g = np.random.default_rng(100 + m)
    W = g.standard_normal((m, 3 * m + 5))
    G = W @ W.T / (3 * m + 5)
    M = np.diag(g.standard_normal(m) * 2.0) + 0.05 * g.standard_normal((m, m))
    M = 0.5 * (M + M.T)
    ref = _woodbury_inner(M, G)
    ld = 80
    Md = torch.zeros(ld, ld, dtype=F64, device=dev)
    Gd = torch.zeros(ld, ld, dtype=F64, device=dev)
    Od = torch.full((ld, ld), float("nan"), dtype=F64, device=dev)
    Md[:m, :m] = torch.from_numpy(M)
    Gd[:m, :m] = torch.from_numpy(G)
    _lib.check(ctx.lib.fh_woodbury_inner(ctx.h, Md.data_ptr(), ld, Gd.data_ptr(), ld, Od.data_ptr(), ld, m, _lib.stream()),
               "fh_woodbury_inner")
    got = Od[:m, :m].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    assert torch.isnan(Od[m:, :]).all() and torch.isnan(Od[:m, m:]).all()  # nothing outside [:m, :m] is touched
    assert ctx.lib.fh_woodbury_inner(ctx.h, Md.data_ptr(), ld, Gd.data_ptr(), ld, Od.data_ptr(), ld, 65, _lib.stream()) < 0
