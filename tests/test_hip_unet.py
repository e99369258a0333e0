"""GPU parity of the hand-written UNet kernels: per-op against plain PyTorch fp32 references, whole network
(forward + input-VJP) against the PyTorch-ROCm backend and against golden vectors from the reference."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.backends.cudnn.allow_tf32 = False
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _lib():
    from free_hunch_amd import _lib as L
    return L, L.load()


@pytest.mark.parametrize("shape", [(1, 64, 64, 128, 128, 3), (1, 16, 16, 256, 96, 3), (2, 8, 8, 64, 256, 3),
                                   (1, 32, 32, 160, 64, 1), (1, 256, 256, 32, 128, 3), (1, 9, 13, 32, 6, 3)])
def test_conv_forward_vs_torch(dev, shape):
    L, lib = _lib()
    N, H, W, Ci, Co, k = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    res = torch.randn(N, Co, H, W, generator=g).to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2) + res.double()
    xn = x.permute(0, 2, 3, 1).contiguous()
    wn = w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci).contiguous()
    rn = res.permute(0, 2, 3, 1).contiguous()
    out = torch.empty(N, H, W, Co, device=dev)
    ks = lib.fh_conv2d_splitk(N, H, W, Ci, Co, k, k)
    ws = torch.empty(max(ks, 1), N * H * W, Co, device=dev)
    L.check(lib.fh_conv2d_nhwc(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), rn.data_ptr(), out.data_ptr(), ws.data_ptr(), ks, N,
                               H, W, Ci, Co, k, k, k // 2, 1, L.stream()), "conv")
    # exact-fp32 MFMA accumulation: error ~ 1e-7 * sum|a b|
    assert rel(out.permute(0, 3, 1, 2), ref) < 5e-6


def test_bgemm_all_transposes(dev):
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    M, N, K, batch = 100, 70, 50, 3
    A = torch.randn(batch, M, K, generator=g).to(dev)
    B = torch.randn(batch, N, K, generator=g).to(dev)
    ref = 0.5 * torch.bmm(A.double(), B.double().transpose(1, 2))
    for ta in (0, 1):
        for tb in (0, 1):
            Am = (A.transpose(1, 2) if ta else A).contiguous()
            Bm = (B.transpose(1, 2) if tb else B).contiguous()
            C = torch.empty(batch, M, N, device=dev)
            L.check(lib.fh_bgemm_f32(Am.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, Am.shape[2], Bm.shape[2], N, ta,
                                     tb, batch, 1, M * K, 0, N * K, 0, M * N, 0, 0.5, L.stream()), "bgemm")
            assert rel(C, ref) < 5e-6, (ta, tb)


@pytest.mark.parametrize("act,ss", [(1, True), (1, False), (0, False)])
def test_groupnorm_fwd_bwd_vs_torch(dev, act, ss):
    L, lib = _lib()
    g = torch.Generator().manual_seed(9)
    N, H, W, C = 2, 12, 10, 96
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dev).requires_grad_()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    e = torch.randn(N, 2 * C, generator=g).to(dev)
    dy = torch.randn(N, C, H, W, generator=g).to(dev)
    t = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if ss:
        t = t * (1 + e[:, :C, None, None]) + e[:, C:, None, None]
    ref = F.silu(t) if act else t
    (gref,) = torch.autograd.grad((ref * dy).sum(), x)
    xn = x.detach().permute(0, 2, 3, 1).contiguous()
    dyn = dy.permute(0, 2, 3, 1).contiguous()
    stats, sums = torch.empty(N, 32, 2, device=dev), torch.empty(N, 32, 2, device=dev)
    y, dx = torch.empty_like(xn), torch.empty_like(xn)
    sc, sh = (e[:, :C], e[:, C:]) if ss else (None, None)
    st = L.stream()
    scratch = torch.empty(lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=dev)
    L.check(lib.fh_groupnorm_stats(xn.data_ptr(), stats.data_ptr(), scratch.data_ptr(), N, H * W, C, st), "stats")
    L.check(lib.fh_groupnorm_apply(xn.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                   sc.data_ptr() if ss else None, sh.data_ptr() if ss else None, 2 * C, y.data_ptr(), N,
                                   H * W, C, act, st), "apply")
    L.check(lib.fh_groupnorm_bwd(xn.data_ptr(), dyn.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 sc.data_ptr() if ss else None, sh.data_ptr() if ss else None, 2 * C, sums.data_ptr(),
                                 scratch.data_ptr(), dx.data_ptr(), N, H * W, C, act, 0, st), "bwd")
    assert rel(y.permute(0, 3, 1, 2), ref.detach()) < 2e-5
    assert rel(dx.permute(0, 3, 1, 2), gref) < 5e-5


def _pair(cfg_o, seed, dev):
    from free_hunch_amd import unet as hu
    kw = {k: getattr(cfg_o, k) for k in ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                                         "attention_resolutions", "num_heads", "num_head_channels",
                                         "use_scale_shift_norm", "resblock_updown", "use_new_attention_order",
                                         "conv_resample")}
    cfg = hu.UNetConfig(**kw)
    sd = hu.seeded_state(cfg, seed)
    nets = []
    for backend in ("hip", "torch"):
        m = hu.UNetModel(cfg, backend=backend)
        m.load_state_dict(sd)
        nets.append(m.to(dev).eval())
    return nets, cfg


NEW_ORDER = inputs.SMALL_A.__class__(**{**inputs.SMALL_A.__dict__, "use_new_attention_order": True,
                                        "use_scale_shift_norm": False})


@pytest.mark.parametrize("which", ["A", "A_new_order_plain_norm", "B_conv_resample", "B_pool_resample"])
def test_unet_hip_vs_torch_backend(dev, which):
    cfg_o = {"A": inputs.SMALL_A, "A_new_order_plain_norm": NEW_ORDER, "B_conv_resample": inputs.SMALL_B,
             "B_pool_resample": inputs.SMALL_B.__class__(**{**inputs.SMALL_B.__dict__, "conv_resample": False})}[which]
    (hip, ref), cfg = _pair(cfg_o, 11, dev)
    x = (inputs.randn((2, 3, 64, 64), 3, torch.float32) * 0.7).to(dev)
    t = torch.tensor([500, 500], device=dev)
    cot = inputs.randn((2, cfg.out_channels, 64, 64), 4, torch.float32).to(dev)
    outs = []
    for m in (hip, ref):
        xi = x.clone().requires_grad_()
        y = m(xi, t)
        (gx,) = torch.autograd.grad((y * cot).sum(), xi)
        outs.append((y.detach(), gx))
    assert rel(outs[0][0], outs[1][0]) < 2e-4
    assert rel(outs[0][1], outs[1][1]) < 5e-4


@pytest.mark.parametrize("tag,cfg_o", [("unet_a", inputs.SMALL_A), ("unet_b", inputs.SMALL_B)])
def test_unet_hip_vs_reference_golden(dev, gold, tag, cfg_o):
    """Raw UNet output, preconditioned denoiser and its input-VJP against vectors produced by the reference's own
    UNetModel / iDDPMLinearPrecond: tests/golden/unet_a.npz (scale-shift norm, legacy attention order, learn_sigma) and
    unet_b.npz (plain norm, new attention order, 3 output channels - the denoiser formed as the reference's wrapper does)."""
    from free_hunch_amd.precond import iDDPMLinearPrecond
    g = gold(tag)
    seed = int(g["seed"])
    (hip, _), cfg = _pair(cfg_o, seed, dev)
    net = iDDPMLinearPrecond(hip, 64, 3).to(dev) if cfg_o.learn_sigma else None
    x = (inputs.randn((1, 3, 64, 64), seed + 100) * 3.0).to(dev)
    for j in range(3):
        sigma = torch.tensor(float(g[f"sigma_{j}"]), dtype=torch.float64, device=dev)
        tstep = torch.from_numpy(g[f"tstep_{j}"]).long().flatten().to(dev)
        c_in = 1 / (sigma ** 2 + 1).sqrt()
        with torch.no_grad():
            raw = hip(c_in.float() * x.float(), tstep)
        ref = torch.from_numpy(g[f"raw_{j}"]).to(dev)
        assert rel(raw, ref) < 5e-4
        xt = x.clone().requires_grad_()
        if net is not None:
            D, var = net(xt, sigma)
        else:
            D = xt.float() - sigma.float() * hip(c_in.float() * xt.float(), tstep)
        Dref = torch.from_numpy(g[f"D_{j}"]).to(dev)
        assert D.dtype == Dref.dtype
        assert float((D - Dref).abs().max()) < 1e-3
        cot = inputs.randn(D.shape, seed + 200 + j).to(D.dtype).to(dev)
        (vjp,) = torch.autograd.grad((cot * D).sum(), xt)
        assert rel(vjp, torch.from_numpy(g[f"vjp_{j}"]).to(dev)) < 2e-3


def wino_weights(w):
    """[Cout][Cin][3][3] -> [4][Cout][3][Cin] F(2,3) transform along kx (see fh_conv3x3_wino_nhwc)."""
    w0, w1, w2 = w[..., 0], w[..., 1], w[..., 2]  # [Co][Ci][ky]
    u = torch.stack([w0, (w0 + w1 + w2) / 2, (w0 - w1 + w2) / 2, w2], 0)  # [4][Co][Ci][ky]
    return u.permute(0, 1, 3, 2).contiguous()


@pytest.mark.parametrize("shape", [(1, 64, 64, 128, 128), (2, 16, 32, 64, 96), (1, 256, 256, 32, 128), (1, 9, 14, 32, 6),
                                   (3, 8, 8, 160, 64)])
def test_conv_winograd_vs_torch(dev, shape):
    L, lib = _lib()
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    res = torch.randn(N, Co, H, W, generator=g).to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    xn = x.permute(0, 2, 3, 1).contiguous()
    rn = res.permute(0, 2, 3, 1).contiguous()
    wu = wino_weights(w)
    out = torch.empty(N, H, W, Co, device=dev)
    L.check(lib.fh_conv3x3_wino_nhwc(xn.data_ptr(), wu.data_ptr(), b.data_ptr(), rn.data_ptr(), out.data_ptr(), N, H, W,
                                     Ci, Co, L.stream()), "wino")
    # F(2,3) has ~2x the error constant of the direct fp32 sum
    assert rel(out.permute(0, 3, 1, 2), ref) < 1e-5


@pytest.mark.parametrize("shape", [(1, 64, 64, 128, 128, 3, 1), (2, 16, 32, 64, 96, 3, 1), (1, 256, 256, 32, 128, 3, 1),
                                   (1, 9, 14, 32, 6, 3, 1), (3, 8, 8, 160, 64, 3, 1), (2, 32, 32, 256, 320, 1, 1),
                                   (1, 33, 31, 64, 130, 3, 2), (8, 64, 64, 128, 256, 3, 1), (2, 128, 128, 64, 160, 3, 1),
                                   (4, 128, 256, 96, 128, 3, 1), (16, 32, 32, 64, 384, 3, 1), (9, 64, 64, 32, 256, 3, 1)])
def test_conv_split_bf16_vs_float64(dev, shape):
    """fh_conv2d_x6_nhwc (exact 3-way bf16 split, six products) against float64: the error must be that of an fp32
    dot product - checked relative to the fp32-MFMA kernel on the same data (<= 1.5x its error, and < 2e-6 of scale)."""
    from free_hunch_amd.unet_hip import _split3
    L, lib = _lib()
    N, H, W, Ci, Co, k, stride = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = (torch.randn(N, Ci, H, W, generator=g) * (0.2 + 3 * torch.rand(N, Ci, H, W, generator=g))).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    pad = k // 2
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad, stride=stride)
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = torch.randn(N, Co, Ho, Wo, generator=g).to(dev)
    ref = ref + res.double()
    xn = x.permute(0, 2, 3, 1).contiguous()
    rn = res.permute(0, 2, 3, 1).contiguous()
    wf = w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci).contiguous()
    wx = _split3(wf)
    planes = wx.float().permute(0, 3, 1, 2, 4).reshape(3, Co, k * k, Ci)
    assert torch.equal(planes.sum(0), wf) or float((planes.double().sum(0) - wf.double()).abs().max()) == 0.0
    outs = []
    for which in ("x6", "f32"):
        out = torch.full((N, Ho, Wo, Co), float("nan"), device=dev)
        if which == "x6":
            L.check(lib.fh_conv2d_x6_nhwc(xn.data_ptr(), wx.data_ptr(), b.data_ptr(), rn.data_ptr(), out.data_ptr(), None,
                                          1, N, H, W, Ci, Co, k, k, pad, stride, L.stream()), "x6")
        else:
            L.check(lib.fh_conv2d_nhwc(xn.data_ptr(), wf.data_ptr(), b.data_ptr(), rn.data_ptr(), out.data_ptr(), None, 1,
                                       N, H, W, Ci, Co, k, k, pad, stride, L.stream()), "f32")
        outs.append(float((out.permute(0, 3, 1, 2).double() - ref).abs().max()))
    scale = float(ref.abs().max())
    assert outs[0] < 2e-6 * scale, (outs, scale)
    assert outs[0] <= 1.5 * outs[1] + 1e-7 * scale, (outs, scale)


@pytest.mark.parametrize("shape", [(2, 64, 64, 128, 6), (1, 256, 256, 128, 3), (3, 33, 47, 64, 8), (1, 16, 16, 32, 1)])
def test_conv_thin_vs_float64(dev, shape):
    """fh_conv3x3_thin_nhwc (Cout <= 8, direct form) against float64."""
    L, lib = _lib()
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape) + 9)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wf = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).contiguous()
    out = torch.full((N, H, W, Co), float("nan"), device=dev)
    L.check(lib.fh_conv3x3_thin_nhwc(xn.data_ptr(), wf.data_ptr(), b.data_ptr(), out.data_ptr(), N, H, W, Ci, Co,
                                     L.stream()), "thin")
    assert rel(out.permute(0, 3, 1, 2), ref) < 2e-6


@pytest.mark.parametrize("shape", [(1, 256, 256, 64, 128, True), (8, 64, 64, 96, 256, False), (16, 32, 32, 64, 384, True)])
def test_conv_with_fused_groupnorm_vs_two_kernel_path(dev, shape):
    """fh_conv2d_x6_norm_nhwc = fh_groupnorm_apply + fh_conv2d_x6_nhwc bit for bit (same affine constants, same SiLU),
    and both within fp32 accuracy of a float64 GroupNorm -> SiLU -> conv."""
    from free_hunch_amd.unet_hip import _split3
    L, lib = _lib()
    N, H, W, Ci, Co, with_ss = shape
    assert lib.fh_conv2d_x6_norm_supported(N, H, W, Ci, Co) == 1
    g = torch.Generator().manual_seed(sum(shape[:5]) + 21)
    x = (torch.randn(N, H, W, Ci, generator=g) * 1.7 + 0.3).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(Ci, generator=g)).to(dev), (0.1 * torch.randn(Ci, generator=g)).to(dev)
    ss = (0.3 * torch.randn(N, 2 * Ci, generator=g)).to(dev) if with_ss else None
    scale, shift = (ss[:, :Ci], ss[:, Ci:]) if with_ss else (None, None)
    res = torch.randn(N, H, W, Co, generator=g).to(dev)
    wx = _split3(w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).contiguous())
    st = L.stream()
    stats = torch.empty(N, 32, 2, device=dev)
    scratch = torch.empty(lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=dev)
    L.check(lib.fh_groupnorm_stats(x.data_ptr(), stats.data_ptr(), scratch.data_ptr(), N, H * W, Ci, st), "stats")
    sp = lambda t: None if t is None else t.data_ptr()
    stride = 0 if ss is None else ss.stride(0)
    y = torch.empty_like(x)
    L.check(lib.fh_groupnorm_apply(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sp(scale), sp(shift),
                                   stride, y.data_ptr(), N, H * W, Ci, 1, st), "apply")
    two = torch.empty(N, H, W, Co, device=dev)
    L.check(lib.fh_conv2d_x6_nhwc(y.data_ptr(), wx.data_ptr(), b.data_ptr(), res.data_ptr(), two.data_ptr(), None, 1, N, H,
                                  W, Ci, Co, 3, 3, 1, 1, st), "conv")
    table = torch.empty(N, 2, Ci, device=dev)
    L.check(lib.fh_groupnorm_table(stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sp(scale), sp(shift), stride,
                                   table.data_ptr(), N, Ci, st), "table")
    one = torch.full((N, H, W, Co), float("nan"), device=dev)
    L.check(lib.fh_conv2d_x6_norm_nhwc(x.data_ptr(), table.data_ptr(), 1, wx.data_ptr(), b.data_ptr(), res.data_ptr(),
                                       one.data_ptr(), N, H, W, Ci, Co, st), "fused")
    assert torch.equal(one, two)
    xd = x.double().permute(0, 3, 1, 2)
    ref = F.group_norm(xd, 32, gamma.double(), beta.double(), eps=1e-5)
    if with_ss:
        ref = ref * (1 + scale.double()[:, :, None, None]) + shift.double()[:, :, None, None]
    ref = F.conv2d(F.silu(ref), w.double(), b.double(), padding=1) + res.double().permute(0, 3, 1, 2)
    assert rel(one.permute(0, 3, 1, 2), ref) < 2e-5
    assert lib.fh_conv2d_x6_norm_supported(1, 16, 16, 64, 128) == 0


def test_conv_split_bf16_splitk(dev):
    """K-split path of the split-bf16 kernel (small grids): partial sums through the workspace + fixed-order reduce."""
    from free_hunch_amd.unet_hip import _split3
    L, lib = _lib()
    N, H, W, Ci, Co = 1, 8, 8, 512, 96
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wx = _split3(w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).contiguous())
    out = torch.empty(N, H, W, Co, device=dev)
    ws = torch.empty(4, N * H * W, Co, device=dev)
    L.check(lib.fh_conv2d_x6_nhwc(xn.data_ptr(), wx.data_ptr(), b.data_ptr(), None, out.data_ptr(), ws.data_ptr(), 4, N,
                                  H, W, Ci, Co, 3, 3, 1, 1, L.stream()), "x6 split-K")
    assert rel(out.permute(0, 3, 1, 2), ref) < 2e-6


def test_unet_ffhq256_hip_vs_torch_backend(dev):
    """The benchmark architecture at full size (exercises the 128x128 / Winograd / split-K kernel choices): forward and
    input-VJP of the HIP backend against the PyTorch-ROCm backend, same seeded weights."""
    from free_hunch_amd import unet as hu
    cfg = hu.FFHQ256
    sd = hu.seeded_state(cfg, 0)
    outs = []
    x = (inputs.randn((1, 3, 256, 256), 8, torch.float32) * 0.5).to(dev)
    t = torch.tensor([300], device=dev)
    cot = inputs.randn((1, 6, 256, 256), 9, torch.float32).to(dev)
    for backend in ("hip", "torch"):
        m = hu.UNetModel(cfg, backend=backend)
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        xi = x.clone().requires_grad_()
        y = m(xi, t)
        (gx,) = torch.autograd.grad((y * cot).sum(), xi)
        outs.append((y.detach(), gx))
        del m
    assert rel(outs[0][0], outs[1][0]) < 5e-4
    assert rel(outs[0][1], outs[1][1]) < 2e-3


def test_unet_imagenet256_hip_vs_torch_backend(dev):
    """The ImageNet-256 architecture (configs[0], [2], [4]: 42 ResBlocks, attention at 32 / 16 / 8 with T up to 1024) at
    full size, batch 2: forward and input-VJP of the HIP backend against the PyTorch-ROCm backend, same seeded weights."""
    from free_hunch_amd import unet as hu
    cfg = hu.IMAGENET256
    sd = hu.seeded_state(cfg, 1)
    outs = []
    x = (inputs.randn((2, 3, 256, 256), 18, torch.float32) * 0.5).to(dev)
    t = torch.tensor([700, 40], device=dev)
    cot = inputs.randn((2, 6, 256, 256), 19, torch.float32).to(dev)
    for backend in ("hip", "torch"):
        m = hu.UNetModel(cfg, backend=backend)
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        xi = x.clone().requires_grad_()
        y = m(xi, t)
        (gx,) = torch.autograd.grad((y * cot).sum(), xi)
        outs.append((y.detach(), gx))
        del m
        torch.cuda.empty_cache()
    assert rel(outs[0][0], outs[1][0]) < 5e-4
    assert rel(outs[0][1], outs[1][1]) < 2e-3


# ---------------------------------------------------------------- f4: reduced-precision (bf16-compute) torso
@pytest.mark.parametrize("shape", [(1, 64, 64, 128, 128, 3, 1), (1, 256, 256, 32, 128, 3, 1), (8, 64, 64, 128, 256, 3, 1),
                                   (3, 8, 8, 160, 64, 3, 1), (2, 32, 32, 256, 320, 1, 1), (4, 128, 256, 96, 128, 3, 1)])
def test_conv_bf16_mode_is_exact_bf16_compute(dev, shape):
    """fh_unet_set_precision(1): the convolution kernels keep only the leading bf16 plane of both operands (one MFMA product
    instead of six).  That must be EXACTLY "round operands to bf16, multiply, accumulate in fp32": against a float64
    convolution of the bf16-rounded operands the error is an fp32 accumulation error (< 2e-6 of scale), while against the
    unrounded float64 result it is the bf16 rounding (1e-3 .. 1e-2) - which also proves the mode was active."""
    from free_hunch_amd.unet_hip import _split3
    L, lib = _lib()
    N, H, W, Ci, Co, k, stride = shape
    g = torch.Generator().manual_seed(sum(shape) + 17)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    pad = k // 2
    exact = F.conv2d(x.double(), w.double(), b.double(), padding=pad, stride=stride)
    rounded = F.conv2d(x.bfloat16().double(), w.bfloat16().double(), b.double(), padding=pad, stride=stride)
    Ho, Wo = exact.shape[2], exact.shape[3]
    xn = x.permute(0, 2, 3, 1).contiguous()
    wx = _split3(w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci).contiguous())
    out = torch.full((N, Ho, Wo, Co), float("nan"), device=dev)
    lib.fh_unet_set_precision(1)
    try:
        L.check(lib.fh_conv2d_x6_nhwc(xn.data_ptr(), wx.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, 1, N, H, W, Ci,
                                      Co, k, k, pad, stride, L.stream()), "x6 bf16")
        torch.cuda.synchronize()
    finally:
        lib.fh_unet_set_precision(0)
    got = out.permute(0, 3, 1, 2).double()
    scale = float(exact.abs().max())
    assert float((got - rounded).abs().max()) < 2e-6 * scale
    e = float((got - exact).abs().max()) / scale
    assert 2e-4 < e < 3e-2, e


@pytest.mark.parametrize("shape", [(1, 64, 64, 128, 128, 3, 1), (8, 64, 64, 128, 256, 3, 1), (2, 32, 32, 256, 320, 1, 1),
                                   (2, 128, 128, 64, 128, 3, 1)])
def test_conv_fp16_mode_is_exact_half_compute(dev, shape):
    """fh_unet_set_precision(3): one plane of half-precision operands on v_mfma_f32_32x32x16_f16.  Against a float64 convolution
    of the half-ROUNDED operands the error is an fp32 accumulation error (< 2e-6 of scale); against the unrounded float64 result
    it is the half rounding (1e-4 .. 3e-3) - which also proves the mode was active."""
    from free_hunch_amd.unet_hip import _half_plane
    L, lib = _lib()
    N, H, W, Ci, Co, k, stride = shape
    g = torch.Generator().manual_seed(sum(shape) + 23)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    pad = k // 2
    exact = F.conv2d(x.double(), w.double(), b.double(), padding=pad, stride=stride)
    rounded = F.conv2d(x.half().double(), w.half().double(), b.double(), padding=pad, stride=stride)
    Ho, Wo = exact.shape[2], exact.shape[3]
    xn = x.permute(0, 2, 3, 1).contiguous()
    wh = _half_plane(w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci).contiguous())
    out = torch.full((N, Ho, Wo, Co), float("nan"), device=dev)
    lib.fh_unet_set_precision(3)
    try:
        L.check(lib.fh_conv2d_x6_nhwc(xn.data_ptr(), wh.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, 1, N, H, W, Ci,
                                      Co, k, k, pad, stride, L.stream()), "x6 fp16")
        torch.cuda.synchronize()
    finally:
        lib.fh_unet_set_precision(0)
    got = out.permute(0, 3, 1, 2).double()
    scale = float(exact.abs().max())
    assert float((got - rounded).abs().max()) < 2e-6 * scale
    e = float((got - exact).abs().max()) / scale
    assert 2e-5 < e < 5e-3, e


def test_unet_bf16_mode_vs_fp32(dev):
    """The FFHQ-256 architecture in the reduced-precision mode (UNetModel(dtype="bf16"), the counterpart of the reference's
    use_fp16 torso, openai_fp16_util.py:15-32) against the fp32-accurate default: forward and input-VJP agree to bf16
    precision (a few 1e-3 .. 1e-2 relative), i.e. NOT to the fp32 parity bar - the mode is reported separately."""
    from free_hunch_amd import unet as hu
    cfg = hu.FFHQ256
    sd = hu.seeded_state(cfg, 0)
    x = (inputs.randn((2, 3, 256, 256), 18, torch.float32) * 0.5).to(dev)
    t = torch.tensor([300, 300], device=dev)
    cot = inputs.randn((2, 6, 256, 256), 19, torch.float32).to(dev)
    outs = []
    for dtype in ("fp32", "bf16", "bf16x3"):
        m = hu.UNetModel(cfg, backend="hip", dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        xi = x.clone().requires_grad_()
        y = m(xi, t)
        (gx,) = torch.autograd.grad((y * cot).sum(), xi)
        outs.append((y.detach(), gx))
        del m
    ey, eg = rel(outs[1][0], outs[0][0]), rel(outs[1][1], outs[0][1])
    assert 1e-5 < ey < 5e-2, ey
    assert 1e-5 < eg < 1e-1, eg
    # "bf16x3" (two bf16 planes per operand, three products): ~ 2^-16 per product - two orders below the bf16 mode, inside the
    # north star's 1e-3 max-abs bar on the network output, but not fp32: reported as its own mode
    ey3, eg3 = rel(outs[2][0], outs[0][0]), rel(outs[2][1], outs[0][1])
    assert 1e-8 < ey3 < 1e-4 and ey3 < ey / 30, (ey3, ey)
    assert 1e-8 < eg3 < 3e-4 and eg3 < eg / 30, (eg3, eg)
    assert float((outs[2][0] - outs[0][0]).abs().max()) < 1e-3
    # and the default is untouched by having run the reduced mode in the same process
    m = hu.UNetModel(cfg, backend="hip")
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    assert rel(m(x, t), outs[0][0]) == 0.0


def test_reduced_precision_mode_vs_reference_fp16_golden(dev, gold):
    """SURVEY 8(f) item 4 against the reference's OWN reduced-precision path: tests/golden/unet_a_fp16.npz holds the raw
    network output, the denoiser and its input-VJP of `create_model(use_fp16=True)` (float16 torso, openai_unet.py:464,
    625-638, 677) on the inputs of unet_a.npz.  The reference's fp16 result sits d_ref from its fp32 result; the HIP
    reduced-precision modes must sit within the same order of their own fp32 result and of the reference's fp16 one:
      * `unet_dtype = fp16` (what `use_fp16 True` selects): convolution operands rounded to IEEE half on the f16 MFMA - the
        reference's torso arithmetic with fp32 storage between layers: within 3 x d_ref of fp32, 4 x d_ref of the reference's
        fp16 output;
      * `unet_dtype = bf16` rounds convolution operands to bfloat16 (8-bit significand, fp32 storage and accumulation) - NOT the
        reference's float16 torso (11-bit significand, fp16 storage): up to 2^3 x the reference's distance per rounding, so
        the bound is 16 x d_ref (measured: see the report), and the distance to the reference's fp16 output obeys the triangle
        inequality with both.  The mode is reported separately from the fp32 headline; it is bf16-compute, not fp16."""
    from free_hunch_amd import unet as hu
    g16, g32 = gold("unet_a_fp16"), gold("unet_a")
    seed = int(g32["seed"])
    cfg = hu.UNetConfig(**{k: getattr(inputs.SMALL_A, k) for k in
                           ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                            "attention_resolutions", "num_heads", "num_head_channels", "use_scale_shift_norm",
                            "resblock_updown", "use_new_attention_order")})
    x = (inputs.randn((1, 3, 64, 64), seed + 100) * 3.0).to(dev)
    rep = {}
    for mode in ("fp16", "bf16"):
        m = hu.UNetModel(cfg, backend="hip", dtype=mode)
        m.load_state_dict(hu.seeded_state(cfg, seed))
        m = m.to(dev).eval()
        for j in range(3):
            sigma = torch.tensor(float(g32[f"sigma_{j}"]), dtype=torch.float64, device=dev)
            tstep = torch.from_numpy(g32[f"tstep_{j}"]).long().flatten().to(dev)
            c_in = 1 / (sigma ** 2 + 1).sqrt()
            with torch.no_grad():
                raw = m(c_in.float() * x.float(), tstep)
            r32, r16 = torch.from_numpy(g32[f"raw_{j}"]).to(dev), torch.from_numpy(g16[f"raw_{j}"]).to(dev)
            d_ref, d_hip, d_x = rel(r16, r32), rel(raw, r32), rel(raw, r16)
            rep[f"{mode}_{j}"] = dict(sigma=float(sigma), ref_fp16_vs_ref_fp32=d_ref, hip_vs_ref_fp32=d_hip, hip_vs_ref_fp16=d_x)
            assert 1e-5 < d_ref < 2e-2, d_ref            # the reference's fp16 torso really differs from its fp32 one
            # fp16: the same operand roundings as the reference's torso (minus its half-precision storage between layers):
            # as far from fp32 as the reference's fp16 output is, and no further from that output than the two are from fp32
            k = 3 if mode == "fp16" else 16
            assert d_hip < k * d_ref, (mode, j, d_hip, d_ref)
            assert d_x < (k + 1) * d_ref, (mode, j, d_x, d_ref)
            assert d_hip > 0.02 * d_ref, (mode, j, d_hip, d_ref)  # (the mode is active)
        del m
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "reduced_precision_report.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(rep, open(path, "w"), indent=1)


# ---------------------------------------------------------------- group-sum epilogue of the split-bf16 convolutions
GN_EPI_SHAPES = [(8, 64, 64, 128, 128, 3), (2, 128, 128, 64, 384, 3), (8, 32, 32, 256, 256, 3), (1, 256, 256, 32, 128, 3),
                 (8, 64, 64, 128, 256, 1), (2, 64, 64, 96, 160, 3)]


@pytest.mark.parametrize("shape", GN_EPI_SHAPES)
@pytest.mark.parametrize("mode", [0, 1])
def test_conv_group_sum_epilogue(dev, shape, mode):
    """fh_conv2d_x6_nhwc_gn: the block partials a convolution leaves for the GroupNorm that consumes its output.
    mode 0 - (mean, rstd) from fh_groupnorm_finalize against fh_groupnorm_stats on the written output; mode 1 - the two
    backward sums against those fh_groupnorm_bwd forms from (x, dy = the output) - both to 2e-6 (fp32 statistics of the same
    numbers, different summation order), over the 256- / 128- / 64-row tile variants, a 1 x 1 convolution and channel
    counts whose groups straddle the 128-column tiles (Cout = 384: 12 channels per group; 160: 5).  Output unchanged."""
    from free_hunch_amd.unet_hip import _split3
    L, lib = _lib()
    N, H, W, Ci, Co, k = shape
    g = torch.Generator().manual_seed(sum(shape) + 31 + mode)
    x = torch.randn(N, H, W, Ci, generator=g).to(dev)
    w = (torch.randn(Co, k * k, Ci, generator=g) / math.sqrt(Ci * k * k)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    res = torch.randn(N, H, W, Co, generator=g).to(dev)
    wx = _split3(w.contiguous())
    pad, P = k // 2, H * W
    chunks = lib.fh_conv2d_x6_gn_chunks(1, N, H, W, Ci, Co, k, k, pad, 1)
    assert chunks > 0, "these shapes take the epilogue-capable launches"
    plain = torch.empty(N, H, W, Co, device=dev)
    L.check(lib.fh_conv2d_x6_nhwc(x.data_ptr(), wx.data_ptr(), b.data_ptr(), res.data_ptr(), plain.data_ptr(), None, 1, N, H,
                                  W, Ci, Co, k, k, pad, 1, L.stream()), "x6")
    partial = torch.full((N * chunks * 64,), float("nan"), dtype=torch.float64, device=dev)
    e = L.FhGnEpilogue()
    e.partial, e.mode, e.act = partial.data_ptr(), mode, 1
    keep = []
    gamma, beta = (1 + 0.1 * torch.randn(Co, generator=g)).to(dev), (0.1 * torch.randn(Co, generator=g)).to(dev)
    scale, shift = (0.2 * torch.randn(N, Co, generator=g)).to(dev), (0.2 * torch.randn(N, Co, generator=g)).to(dev)
    if mode == 1:
        xin = torch.randn(N, H, W, Co, generator=g).to(dev)  # forward input of the GroupNorm whose dL/dy is the conv output
        stats = torch.empty(N, 32, 2, device=dev)
        scr = torch.empty(lib.fh_groupnorm_scratch_doubles(N, P), dtype=torch.float64, device=dev)
        L.check(lib.fh_groupnorm_stats(xin.data_ptr(), stats.data_ptr(), scr.data_ptr(), N, P, Co, L.stream()), "stats")
        table = torch.empty(N, 5, Co, device=dev)
        L.check(lib.fh_groupnorm_bwd_table(stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(),
                                           shift.data_ptr(), Co, table.data_ptr(), N, Co, L.stream()), "table")
        e.x, e.tab = xin.data_ptr(), table.data_ptr()
        keep += [xin, stats, table]
    out = torch.empty(N, H, W, Co, device=dev)
    import ctypes as C
    L.check(lib.fh_conv2d_x6_nhwc_gn(x.data_ptr(), wx.data_ptr(), b.data_ptr(), res.data_ptr(), out.data_ptr(), None, 1, N, H, W,
                                     Ci, Co, k, k, pad, 1, C.byref(e), L.stream()), "x6 gn")
    assert torch.equal(out, plain)
    got = torch.empty(N, 32, 2, device=dev)
    L.check(lib.fh_groupnorm_finalize(partial.data_ptr(), got.data_ptr(), N, chunks, float(P * (Co // 32)), mode, L.stream()), "fin")
    scr = torch.empty(lib.fh_groupnorm_scratch_doubles(N, P), dtype=torch.float64, device=dev)
    if mode == 0:
        want = torch.empty(N, 32, 2, device=dev)
        L.check(lib.fh_groupnorm_stats(out.data_ptr(), want.data_ptr(), scr.data_ptr(), N, P, Co, L.stream()), "stats")
    else:
        want, dx = torch.empty(N, 32, 2, device=dev), torch.empty_like(out)
        L.check(lib.fh_groupnorm_bwd(xin.data_ptr(), out.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                     scale.data_ptr(), shift.data_ptr(), Co, want.data_ptr(), scr.data_ptr(), dx.data_ptr(), N, P,
                                     Co, 1, 0, L.stream()), "gn_bwd")
        dx2 = torch.empty_like(out)
        L.check(lib.fh_groupnorm_bwd_apply(xin.data_ptr(), out.data_ptr(), stats.data_ptr(), got.data_ptr(), gamma.data_ptr(),
                                           beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), Co, dx2.data_ptr(), N, P, Co, 1, 0,
                                           L.stream()), "gn_bwd_apply")
        assert rel(dx2, dx) < 1e-5
    assert float((got - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max())), (got - want).abs().max()


def test_conv_group_sum_epilogue_fused_norm_input(dev):
    """The same epilogue on the fused GroupNorm-input convolution (fh_conv2d_x6_norm_nhwc_gn), 256-row tiles."""
    from free_hunch_amd.unet_hip import _split3
    import ctypes as C
    L, lib = _lib()
    N, H, W, Ci, Co = 2, 256, 256, 64, 128
    g = torch.Generator().manual_seed(404)
    x = torch.randn(N, H, W, Ci, generator=g).to(dev)
    w = (torch.randn(Co, 9, Ci, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    wx = _split3(w.contiguous())
    assert lib.fh_conv2d_x6_norm_supported(N, H, W, Ci, Co)
    table = torch.cat([1 + 0.1 * torch.randn(N, 1, Ci, generator=g), 0.1 * torch.randn(N, 1, Ci, generator=g)], 1).to(dev).contiguous()
    chunks = lib.fh_conv2d_x6_gn_chunks(1, N, H, W, Ci, Co, 3, 3, 1, 1)
    assert chunks == (H * W // 256)
    partial = torch.full((N * chunks * 64,), float("nan"), dtype=torch.float64, device=dev)
    e = L.FhGnEpilogue()
    e.partial, e.mode, e.act = partial.data_ptr(), 0, 0
    out, plain = torch.empty(N, H, W, Co, device=dev), torch.empty(N, H, W, Co, device=dev)
    L.check(lib.fh_conv2d_x6_norm_nhwc(x.data_ptr(), table.data_ptr(), 1, wx.data_ptr(), b.data_ptr(), None, plain.data_ptr(), N,
                                       H, W, Ci, Co, L.stream()), "norm")
    L.check(lib.fh_conv2d_x6_norm_nhwc_gn(x.data_ptr(), table.data_ptr(), 1, wx.data_ptr(), b.data_ptr(), None, out.data_ptr(), N,
                                          H, W, Ci, Co, C.byref(e), L.stream()), "norm gn")
    assert torch.equal(out, plain)
    got, want = torch.empty(N, 32, 2, device=dev), torch.empty(N, 32, 2, device=dev)
    L.check(lib.fh_groupnorm_finalize(partial.data_ptr(), got.data_ptr(), N, chunks, float(H * W * (Co // 32)), 0, L.stream()), "fin")
    scr = torch.empty(lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=dev)
    L.check(lib.fh_groupnorm_stats(out.data_ptr(), want.data_ptr(), scr.data_ptr(), N, H * W, Co, L.stream()), "stats")
    assert float((got - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))


# ---------------------------------------------------------------- half-split mode (fh_unet_set_precision(4), unet_dtype "fp16x3")
def _half_split_conv(L, lib, xn, planes, b, rn, out, ws, ks, shape, amax_of=None):
    """one fh_conv2d_x6_nhwc_gn launch in precision mode 4 (the input's magnitude through fh_gn_epilogue.in_amax)"""
    import ctypes as C
    N, H, W, Ci, Co, k, stride = shape
    amax = torch.empty(16, device=xn.device)  # FH_AMAX_SLOTS
    src = xn if amax_of is None else amax_of
    L.check(lib.fh_absmax_f32(src.data_ptr(), src.numel(), amax.data_ptr(), L.stream()), "absmax")
    e = L.FhGnEpilogue()
    e.partial, e.in_amax = None, amax.data_ptr()
    assert lib.fh_unet_set_precision(4) == 0
    try:
        L.check(lib.fh_conv2d_x6_nhwc_gn(xn.data_ptr(), planes.data_ptr(), None if b is None else b.data_ptr(),
                                         None if rn is None else rn.data_ptr(), out.data_ptr(),
                                         None if ws is None else ws.data_ptr(), ks, N, H, W, Ci, Co, k, k, k // 2, stride,
                                         C.byref(e), L.stream()), "half-split")
    finally:
        lib.fh_unet_set_precision(0)
    return amax


HALF_SPLIT_SHAPES = [(1, 64, 64, 128, 128, 3, 1), (8, 64, 64, 128, 256, 3, 1), (1, 256, 256, 32, 128, 3, 1),
                     (2, 128, 128, 64, 160, 3, 1), (16, 32, 32, 64, 384, 3, 1), (3, 8, 8, 160, 64, 3, 1),
                     (2, 32, 32, 256, 320, 1, 1), (1, 33, 31, 64, 130, 3, 2), (4, 128, 256, 96, 128, 3, 1)]


@pytest.mark.parametrize("mag", [1.0, 3e-7, 2e5])
@pytest.mark.parametrize("shape", HALF_SPLIT_SHAPES)
def test_conv_half_split_vs_float64(dev, shape, mag):
    """Precision mode 4 (two half-precision planes per operand, three products on the f16 MFMA) against float64 - held to
    the SAME bar as the exact bf16 split (test_conv_split_bf16_vs_float64): the error of an fp32 dot product, i.e. <= 1.5 x
    that of the fp32-MFMA kernel on the same data and < 2e-6 of the output scale.  `mag` scales the input by 2e5 / 3e-7: the
    power-of-two activation scale from fh_absmax_f32 keeps the planes inside the half-precision range (an input-gradient
    tensor can have any magnitude)."""
    from free_hunch_amd.unet_hip import _half_split_planes
    L, lib = _lib()
    N, H, W, Ci, Co, k, stride = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = (torch.randn(N, Ci, H, W, generator=g) * (0.2 + 3 * torch.rand(N, Ci, H, W, generator=g)) * mag).to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).to(dev)
    b = (torch.randn(Co, generator=g) * mag).to(dev)
    pad = k // 2
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad, stride=stride)
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = (torch.randn(N, Co, Ho, Wo, generator=g) * mag).to(dev)
    ref = ref + res.double()
    xn = x.permute(0, 2, 3, 1).contiguous()
    rn = res.permute(0, 2, 3, 1).contiguous()
    wf = w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci).contiguous()
    planes = _half_split_planes(wf)
    # the planes reproduce the weights to one fp32 ulp
    body = planes[:-2].view(torch.float16).reshape(2, k * k, Ci // 32, Co, 32).float()
    winv = float(planes[-2:].view(torch.float32))
    back = (body.double().sum(0) * winv).permute(2, 0, 1, 3).reshape(Co, k * k, Ci)
    # (one fp32 ulp; weights below 2^-16 of the largest leave m subnormal: absolute 2^-25 of the scaled unit instead)
    assert bool(((back - wf.double()).abs() <= 2.0 ** -23 * wf.double().abs() + 2.0 ** -25 * winv).all())
    errs = []
    for which in ("half-split", "f32"):
        out = torch.full((N, Ho, Wo, Co), float("nan"), device=dev)
        if which == "f32":
            L.check(lib.fh_conv2d_nhwc(xn.data_ptr(), wf.data_ptr(), b.data_ptr(), rn.data_ptr(), out.data_ptr(), None, 1,
                                       N, H, W, Ci, Co, k, k, pad, stride, L.stream()), "f32")
        else:
            _half_split_conv(L, lib, xn, planes, b, rn, out, None, 1, shape)
        errs.append(float((out.permute(0, 3, 1, 2).double() - ref).abs().max()))
    scale = float(ref.abs().max())
    assert errs[0] < 2e-6 * scale, (errs, scale)
    assert errs[0] <= 1.5 * errs[1] + 1e-7 * scale, (errs, scale)


def test_conv_half_split_splitk_and_missing_magnitude(dev):
    """K-split launch of mode 4 (partials are un-scaled before they reach the workspace), and the error path: a raw-tensor
    launch without the input's magnitude is refused."""
    from free_hunch_amd.unet_hip import _half_split_planes
    L, lib = _lib()
    shape = (1, 8, 8, 512, 96, 3, 1)
    N, H, W, Ci, Co, k, stride = shape
    g = torch.Generator().manual_seed(78)
    x = (torch.randn(N, Ci, H, W, generator=g) * 40).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    planes = _half_split_planes(w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).contiguous())
    out = torch.empty(N, H, W, Co, device=dev)
    ws = torch.empty(4, N * H * W, Co, device=dev)
    amax = _half_split_conv(L, lib, xn, planes, b, None, out, ws, 4, shape)
    assert float(amax.max()) == float(x.abs().max())
    assert rel(out.permute(0, 3, 1, 2), ref) < 2e-6
    assert lib.fh_unet_set_precision(4) == 0
    try:
        rc = lib.fh_conv2d_x6_nhwc(xn.data_ptr(), planes.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, 1, N, H, W,
                                   Ci, Co, 3, 3, 1, 1, L.stream())
    finally:
        lib.fh_unet_set_precision(0)
    assert rc == -1
    assert lib.fh_unet_set_precision(5) == -1


@pytest.mark.parametrize("shape", [(1, 256, 256, 64, 128, True), (8, 64, 64, 96, 256, False), (16, 32, 32, 64, 384, True)])
def test_conv_half_split_fused_groupnorm_input(dev, shape):
    """Mode 4 of fh_conv2d_x6_norm_nhwc (GroupNorm + SiLU applied while the tile is staged, fixed activation scale 2^4):
    no further from a float64 GroupNorm -> SiLU -> conv than the exact split is (both carry the fp32 normalisation's
    rounding), and the two agree to fp32 rounding."""
    from free_hunch_amd.unet_hip import _half_split_planes, _split3
    L, lib = _lib()
    N, H, W, Ci, Co, with_ss = shape
    g = torch.Generator().manual_seed(sum(shape[:5]) + 23)
    x = (torch.randn(N, H, W, Ci, generator=g) * 1.7 + 0.3).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(Ci, generator=g)).to(dev), (0.1 * torch.randn(Ci, generator=g)).to(dev)
    ss = (0.3 * torch.randn(N, 2 * Ci, generator=g)).to(dev) if with_ss else None
    scale, shift = (ss[:, :Ci], ss[:, Ci:]) if with_ss else (None, None)
    res = torch.randn(N, H, W, Co, generator=g).to(dev)
    wf = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).contiguous()
    st = L.stream()
    stats = torch.empty(N, 32, 2, device=dev)
    scratch = torch.empty(lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=dev)
    L.check(lib.fh_groupnorm_stats(x.data_ptr(), stats.data_ptr(), scratch.data_ptr(), N, H * W, Ci, st), "stats")
    sp = lambda t: None if t is None else t.data_ptr()
    stride = 0 if ss is None else ss.stride(0)
    table = torch.empty(N, 2, Ci, device=dev)
    L.check(lib.fh_groupnorm_table(stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sp(scale), sp(shift), stride,
                                   table.data_ptr(), N, Ci, st), "table")
    outs = {}
    for mode, planes in ((0, _split3(wf)), (4, _half_split_planes(wf))):
        out = torch.full((N, H, W, Co), float("nan"), device=dev)
        assert lib.fh_unet_set_precision(mode) == 0
        try:
            L.check(lib.fh_conv2d_x6_norm_nhwc(x.data_ptr(), table.data_ptr(), 1, planes.data_ptr(), b.data_ptr(),
                                               res.data_ptr(), out.data_ptr(), N, H, W, Ci, Co, st), "fused")
        finally:
            lib.fh_unet_set_precision(0)
        outs[mode] = out.permute(0, 3, 1, 2)
    xd = x.double().permute(0, 3, 1, 2)
    ref = F.group_norm(xd, 32, gamma.double(), beta.double(), eps=1e-5)
    if with_ss:
        ref = ref * (1 + scale.double()[:, :, None, None]) + shift.double()[:, :, None, None]
    ref = F.conv2d(F.silu(ref), w.double(), b.double(), padding=1) + res.double().permute(0, 3, 1, 2)
    e0, e4 = rel(outs[0], ref), rel(outs[4], ref)
    assert e4 < 2e-5 and e4 <= 1.25 * e0 + 2e-7, (e0, e4)
    assert rel(outs[4], outs[0]) < 1e-6


def test_half_split_mode_network_accuracy_vs_float64(dev, gold):
    """unet_dtype "fp16x3" at network level.  The reference's float32 UNet (tests/golden/unet_a.npz) is itself only an
    fp32 evaluation; the yardstick is the float64 evaluation of the same network (the PyTorch backend in double).  Asserted
    for raw output and input-VJP at the three recorded noise levels: the half-split mode is no further from float64 than
    1.25 x the exact-split default (+ 1e-7), both are as close to float64 as the reference's own float32 result is (x 1.5),
    and the mode passes test_unet_hip_vs_reference_golden's bounds unchanged."""
    from free_hunch_amd import unet as hu
    from free_hunch_amd.precond import iDDPMLinearPrecond
    g = gold("unet_a")
    seed = int(g["seed"])
    cfg = hu.UNetConfig(**{k: getattr(inputs.SMALL_A, k) for k in
                           ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                            "attention_resolutions", "num_heads", "num_head_channels", "use_scale_shift_norm",
                            "resblock_updown", "use_new_attention_order")})
    sd = hu.seeded_state(cfg, seed)
    x = (inputs.randn((1, 3, 64, 64), seed + 100) * 3.0).to(dev)
    from unittest import mock
    from oracle import unet_oracle as uo
    sd64 = {k: v.double() for k, v in uo.seeded_state(inputs.SMALL_A, seed).items()}

    def m64(xi, tstep):  # the oracle's network with every `.float()` of its GroupNorm32 / softmax made a `.double()` (CPU)
        with mock.patch.object(torch.Tensor, "float", lambda self: self.double()):
            return uo.unet_forward(sd64, inputs.SMALL_A, xi, tstep)
    nets = {}
    for mode in ("fp32", "fp16x3"):
        m = hu.UNetModel(cfg, backend="hip", dtype=mode)
        m.load_state_dict(sd)
        nets[mode] = m.to(dev).eval()
    rep = {}
    for j in range(3):
        sigma = torch.tensor(float(g[f"sigma_{j}"]), dtype=torch.float64, device=dev)
        tstep = torch.from_numpy(g[f"tstep_{j}"]).long().flatten().to(dev)
        c_in = 1 / (sigma ** 2 + 1).sqrt()
        cot = inputs.randn((1, cfg.out_channels, 64, 64), seed + 300 + j).to(dev)
        xi = (c_in * x.double()).float().double().cpu().requires_grad_()   # (the fp32 networks see the fp32 product)
        y64 = m64(xi, tstep.cpu())
        assert y64.dtype == torch.float64
        (g64,) = torch.autograd.grad((y64 * cot.double().cpu()).sum(), xi)
        y64, g64 = y64.detach().to(dev), g64.to(dev)
        ref32 = torch.from_numpy(g[f"raw_{j}"]).to(dev)
        res = {}
        for mode, m in nets.items():
            xi = (c_in.float() * x.float()).requires_grad_()
            y = m(xi, tstep)
            (gx,) = torch.autograd.grad((y * cot).sum(), xi)
            res[mode] = (rel(y.detach(), y64.detach()), rel(gx, g64))
            if mode == "fp16x3":
                assert rel(y.detach(), ref32) < 5e-4
                net = iDDPMLinearPrecond(m, 64, 3).to(dev)
                xt = x.clone().requires_grad_()
                D, _ = net(xt, sigma)
                assert float((D - torch.from_numpy(g[f"D_{j}"]).to(dev)).abs().max()) < 1e-3
                c2 = inputs.randn(D.shape, seed + 200 + j).to(D.dtype).to(dev)
                (vjp,) = torch.autograd.grad((c2 * D).sum(), xt)
                assert rel(vjp, torch.from_numpy(g[f"vjp_{j}"]).to(dev)) < 2e-3
        e_ref = rel(ref32, y64.detach())
        rep[j] = dict(sigma=float(sigma), fwd=dict(exact=res["fp32"][0], half_split=res["fp16x3"][0], reference_fp32=e_ref),
                      vjp=dict(exact=res["fp32"][1], half_split=res["fp16x3"][1]))
        assert res["fp16x3"][0] <= 1.25 * res["fp32"][0] + 1e-7, rep[j]
        assert res["fp16x3"][1] <= 1.25 * res["fp32"][1] + 1e-7, rep[j]
        assert res["fp16x3"][0] <= 1.5 * e_ref + 1e-7, rep[j]
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "half_split_report.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(rep, open(path, "w"), indent=1)


@pytest.mark.parametrize("split", [0, 64])
def test_groupnorm_backward_apply_reports_output_magnitudes(dev, split):
    """fh_groupnorm_bwd_apply_ex(amax2): the pass that writes a gradient tensor also leaves max |dx| (and max |dx2| of a split
    result) - exactly fh_absmax_f32 of what it wrote - for the half-split convolution that reads the tensor next."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(31 + split)
    N, H, W, C = 2, 33, 20, 160
    xn = (torch.randn(N, H, W, C, generator=g) * 2 + 0.5).to(dev)
    dyn = (torch.randn(N, H, W, C, generator=g) * 3e-4).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    acc = (torch.randn(N, H, W, C, generator=g) * 1e-4).to(dev)
    st = L.stream()
    stats, sums = torch.empty(N, 32, 2, device=dev), torch.empty(N, 32, 2, device=dev)
    scratch = torch.empty(lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=dev)
    L.check(lib.fh_groupnorm_stats(xn.data_ptr(), stats.data_ptr(), scratch.data_ptr(), N, H * W, C, st), "stats")
    L.check(lib.fh_groupnorm_bwd_sums(xn.data_ptr(), dyn.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None,
                                      None, 0, sums.data_ptr(), scratch.data_ptr(), N, H * W, C, 1, st), "sums")
    outs = []
    for with_amax in (False, True):
        dx = torch.full((N, H, W, split if split else C), float("nan"), device=dev)
        dx2 = torch.full((N, H, W, C - split), float("nan"), device=dev) if split else None
        amax = torch.zeros(2, 16, device=dev)  # [dx, dx2][FH_AMAX_SLOTS]
        L.check(lib.fh_groupnorm_bwd_apply_ex(xn.data_ptr(), dyn.data_ptr(), stats.data_ptr(), sums.data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), None, None, 0, acc.data_ptr(), None, dx.data_ptr(),
                                              None if dx2 is None else dx2.data_ptr(), split, N, H * W, C, 1,
                                              amax.data_ptr() if with_amax else None, st), "apply_ex")
        outs.append((dx, dx2, amax))
    assert torch.equal(outs[0][0], outs[1][0]) and (split == 0 or torch.equal(outs[0][1], outs[1][1]))
    dx, dx2, amax = outs[1]
    assert float(amax[0].max()) == float(dx.abs().max()) > 0
    assert float(amax[1].max()) == (float(dx2.abs().max()) if split else 0.0)
    one = torch.empty(16, device=dev)
    L.check(lib.fh_absmax_f32(dx.data_ptr(), dx.numel(), one.data_ptr(), st), "absmax")
    assert float(one.max()) == float(amax[0].max())
    assert float(outs[0][2].abs().max()) == 0.0


# ---------------------------------------------------------------- fused attention (fh_attention_fwd / fh_attention_bwd)
def _attention_float64(qkv, heads, new_order):
    """QKVAttentionLegacy / QKVAttention of the reference (openai_unet.py:337-354 / :370-384) in float64 on [N][T][3C] tokens."""
    N, T, C3 = qkv.shape
    C = C3 // 3
    ch = C // heads
    x = qkv.transpose(1, 2)  # [N][3C][T] as the reference's conv1d output
    if new_order:
        q, k, v = x.chunk(3, dim=1)
        q, k, v = (t.reshape(N * heads, ch, T) for t in (q, k, v))
    else:
        q, k, v = x.reshape(N * heads, ch * 3, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(N, C, T)
    return a.transpose(1, 2)  # [N][T][C]


@pytest.mark.parametrize("shape", [(2, 64, 64, 2, 0), (1, 256, 128, 2, 0), (2, 1024, 128, 2, 1), (3, 96, 128, 4, 1),
                                   (1, 1024, 256, 4, 0), (2, 32, 64, 1, 0)])
def test_fused_attention_vs_float64(dev, shape):
    """One-kernel attention and its recomputing backward against the reference's two attention orders in float64: output,
    dq / dk / dv within fp32 rounding (5e-6 of scale; the three-kernel path with materialised weights is held to the same),
    and the two paths agree with each other to that order.  T = 96 (not a multiple of 128) exercises idle waves."""
    L, lib = _lib()
    N, T, C, heads, new_order = shape
    assert lib.fh_attention_supported(T, C, heads) == 1
    g = torch.Generator().manual_seed(sum(shape) + 3)
    qkv = (torch.randn(N, T, 3 * C, generator=g) * 1.5).to(dev)
    dout = torch.randn(N, T, C, generator=g).to(dev)
    q64 = qkv.double().requires_grad_()
    ref = _attention_float64(q64, heads, new_order)
    (gref,) = torch.autograd.grad((ref * dout.double()).sum(), q64)
    out = torch.full((N, T, C), float("nan"), device=dev)
    lse = torch.full((N * heads, T), float("nan"), device=dev)
    st = L.stream()
    L.check(lib.fh_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), N, T, C, heads, new_order, st), "attn fwd")
    assert rel(out, ref.detach()) < 5e-6
    dqkv = torch.full((N, T, 3 * C), float("nan"), device=dev)
    dsum = torch.empty_like(lse)
    L.check(lib.fh_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dsum.data_ptr(),
                                 dqkv.data_ptr(), N, T, C, heads, new_order, st), "attn bwd")
    assert rel(dqkv, gref) < 5e-6
    assert rel(dsum.reshape(N, heads, T), (ref.detach() * dout.double()).reshape(N, T, heads, C // heads).sum(-1).transpose(1, 2)) < 5e-6
    # deterministic: a second run gives the same bits
    dq2 = torch.empty_like(dqkv)
    L.check(lib.fh_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dsum.data_ptr(),
                                 dq2.data_ptr(), N, T, C, heads, new_order, st), "attn bwd")
    assert torch.equal(dq2, dqkv)


def test_fused_attention_unsupported_shapes_fall_back(dev):
    L, lib = _lib()
    assert lib.fh_attention_supported(48, 64, 2) == 0      # T % 32 != 0
    assert lib.fh_attention_supported(64, 96, 2) == 0      # head width 48
    assert lib.fh_attention_supported(64, 256, 2) == 0     # head width 128
    x = torch.zeros(1, 48, 192, device=dev)
    o, l = torch.zeros(1, 48, 64, device=dev), torch.zeros(2, 48, device=dev)
    assert lib.fh_attention_fwd(x.data_ptr(), o.data_ptr(), l.data_ptr(), 1, 48, 64, 2, 0, L.stream()) == -2


def test_unet_fused_attention_equals_three_kernel_path(dev):
    """Network level: forward and input-VJP with the fused attention (default) against the three-kernel path that keeps the
    T x T weights on the tape (FH_ATTN_FUSED=0), legacy and new attention order."""
    for cfg_o in (inputs.SMALL_A, NEW_ORDER):
        (hip, _), cfg = _pair(cfg_o, 13, dev)
        x = (inputs.randn((2, 3, 64, 64), 5, torch.float32) * 0.7).to(dev)
        t = torch.tensor([400, 400], device=dev)
        cot = inputs.randn((2, cfg.out_channels, 64, 64), 6, torch.float32).to(dev)
        outs = []
        for fused in ("1", "0"):
            os.environ["FH_ATTN_FUSED"] = fused
            try:
                xi = x.clone().requires_grad_()
                y = hip(xi, t)
                (gx,) = torch.autograd.grad((y * cot).sum(), xi)
            finally:
                os.environ.pop("FH_ATTN_FUSED", None)
            outs.append((y.detach(), gx))
        assert rel(outs[0][0], outs[1][0]) < 2e-5
        assert rel(outs[0][1], outs[1][1]) < 5e-5
