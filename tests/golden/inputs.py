"""Seeded synthetic INPUTS shared by make_golden.py (which feeds them to the reference) and the tests
(which feed them to the oracle and to the HIP path).  Pure torch-CPU / numpy generators: the same torch
build gives the same stream here and on the GPU box, so large inputs are regenerated instead of stored."""
import numpy as np
import torch

from oracle.unet_oracle import UNetConfig

F64 = torch.float64


def rng(seed):
    return torch.Generator().manual_seed(seed)


def randn(shape, seed, dtype=F64):
    return torch.randn(shape, generator=rng(seed), dtype=dtype)



SMALL_A = UNetConfig(image_size=64, num_channels=32, num_res_blocks=1, channel_mult=(), learn_sigma=True,
                     attention_resolutions="16,8", num_heads=4, num_head_channels=32,
                     use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False)
SMALL_B = UNetConfig(image_size=64, num_channels=32, num_res_blocks=2, channel_mult=(1, 2, 2), learn_sigma=False,
                     attention_resolutions="32", num_heads=2, num_head_channels=-1,
                     use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=True)

# 256x256 with the ImageNet-256 block structure (6 levels, attention at 32/16/8 -> T = 1024 / 256 / 64) at 32 base channels
SMALL_C = UNetConfig(image_size=256, num_channels=32, num_res_blocks=1, channel_mult=(), learn_sigma=True,
                     attention_resolutions="32,16,8", num_heads=4, num_head_channels=32,
                     use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False)


# ---- denoisers for which the reference reproduces itself across hosts (tests/golden/make_golden.py: *_tight, tmpd_*_pos)
GAUSS_PRIOR_VAR = 0.25
DAMP = 0.05


def gauss_prior_denoise(x, sigma, var=GAUSS_PRIOR_VAR):
    """Posterior mean of a N(0, var I) prior at noise level sigma: linear in x, contractive, no clamp."""
    return x * (var / (var + sigma ** 2))


def pp_prior_var(size, seed=77):
    """A smooth, strictly positive per-pixel prior variance in [0.05, 0.5] (for the per-pixel-variance plugins)."""
    return 0.05 + 0.45 * (smooth_image(size, seed).double() + 1.0) / 2.0


def pp_gauss_prior_denoise(x, sigma, var):
    """Posterior mean of independent N(0, var[p]) priors: diagonal Jacobian var / (var + sigma^2) > 0, so TMPD's variance
    field (Jacobian row sums x sigma^2) is positive and spatially varying."""
    return x * (var / (var + sigma ** 2))


def damped_state(seeded_state, cfg, seed, damp=DAMP):
    """`seeded_state` weights with the last convolution (out.2) scaled by `damp`: F = UNet(c_in x) stays small, so the
    denoiser D = clamp(x - sigma F) has a Jacobian close to the clamp mask and the sampler does not amplify the 1e-5
    differences between two fp32 UNet implementations (a random-weight UNet at full scale does)."""
    sd = seeded_state(cfg, seed)
    for k in ("out.2.weight", "out.2.bias"):
        sd[k] = sd[k] * damp
    return sd


def script(seed, shape, n_steps, sig0, neg_gamma_at=None, sig_end=0.5):
    """A scripted alternation of time and space updates with seeded vectors (mimics the Heun call pattern)."""
    g = rng(seed)
    sig = np.geomspace(sig0, sig_end, n_steps + 1)
    steps = []
    for i in range(n_steps):
        x = torch.randn(shape, generator=g, dtype=F64) * sig[i]
        score = -x / sig[i] ** 2 * torch.rand(shape, generator=g, dtype=F64)
        steps.append(("time", dict(x=x, sigma=float(sig[i]), sigma_next=float(sig[i + 1]), score=score)))
        xn = x + 0.3 * sig[i + 1] * torch.randn(shape, generator=g, dtype=F64)
        m0 = 0.5 * x
        dm = 0.4 * (xn - x) + 0.05 * torch.randn(shape, generator=g, dtype=F64)
        if neg_gamma_at == i:
            dm = -dm
        steps.append(("space", dict(x=x, xn=xn, m0=m0, m1=m0 + dm, sigma=float(sig[i + 1]))))
    return steps



def smooth_image(size, seed):
    """Natural-ish test image in [-1,1]: low-pass filtered noise."""
    g = np.random.default_rng(seed)
    f = g.standard_normal((3, size, size))
    fy = np.fft.fftfreq(size)[:, None]
    fx = np.fft.fftfreq(size)[None, :]
    spec = np.fft.fft2(f) / (1 + 40 * np.sqrt(fy ** 2 + fx ** 2)) ** 1.5
    img = np.real(np.fft.ifft2(spec))
    img = img / np.abs(img).max()
    return torch.from_numpy(img[None].astype(np.float32))




def solver256_measurement(name, x, mask=None, noise_seed=131):
    """The measurement fed to the full-size solver fixtures (solver256.npz): built WITHOUT the operator, so that every
    side regenerates it bit for bit - the sharp image itself (decimated for SR, masked for inpainting) plus 0.1 noise."""
    y = x[..., ::4, ::4].clone() if name == "super_resolution" else x.clone()
    y = y + 0.1 * randn(y.shape, noise_seed, torch.float32)
    return y * mask if mask is not None else y


def dense_case(seed, bs, d, n_steps=3):
    """Seeded inputs for the dense-helper chain (online_update_bfgs.py:377-463): a Gaussian prior N(0, S) with
    S = Q diag(lam) Q^T, Q = I - 2 w w^T, gives the four matrices at sigma_0 in closed form (no inverse needed);
    the chain then alternates update_covariance and update_bfgs as `dense_chain` below prescribes."""
    g = rng(seed)
    sig = [5.0, 3.0, 2.0, 1.2, 0.8, 0.5][:n_steps + 1]
    w = torch.randn(bs, d, generator=g, dtype=F64)
    w = w / w.norm(dim=-1, keepdim=True)
    lam = 0.2 + 1.8 * torch.rand(bs, d, generator=g, dtype=F64)

    def qdq(diag):  # (I - 2ww^T) D (I - 2ww^T) in O(d^2)
        dw = diag * w
        wdw = (w * dw).sum(-1)
        return (torch.diag_embed(diag) - 2 * w[:, :, None] * dw[:, None, :] - 2 * dw[:, :, None] * w[:, None, :]
                + 4 * wdw[:, None, None] * w[:, :, None] * w[:, None, :])

    s0 = sig[0]
    mats = dict(C=qdq(1 / (1 / lam + s0 ** -2)), Ci=qdq(1 / lam + s0 ** -2), H=qdq(-1 / (lam + s0 ** 2)),
                Hi=qdq(-(lam + s0 ** 2)))
    x = torch.randn(bs, d, generator=g, dtype=F64) * s0
    score = torch.randn(bs, d, generator=g, dtype=F64) / s0
    e1 = [torch.randn(bs, d, generator=g, dtype=F64) for _ in range(n_steps)]
    e2 = [torch.randn(bs, d, generator=g, dtype=F64) for _ in range(n_steps)]
    return dict(sig=sig, x=x, score=score, e1=e1, e2=e2, **mats)


def dense_chain(case, time_update, space_update, to=lambda t: t):
    """Drive `time_update` (signature of update_covariance) and `space_update` (update_bfgs) through the scripted
    chain; `to` moves inputs to the implementation's device.  Yields the state after every update."""
    sig = case["sig"]
    C, Ci, H, Hi = (to(case[k].clone()) for k in ("C", "Ci", "H", "Hi"))
    x, score = to(case["x"]), to(case["score"])
    mean = x + sig[0] ** 2 * score
    sched = lambda t: t  # noqa: E731  (noise level = time)
    for i in range(len(case["e1"])):
        C, Ci, H, Hi, score, mean = time_update(x, C, Ci, H, Hi, score, mean, sched, sig[i], sig[i + 1])
        yield ("time", i, C, Ci, H, Hi, score, mean)
        xn = x + 0.3 * to(case["e1"][i])
        m1 = mean + 0.4 * (xn - x) + 0.02 * to(case["e2"][i])
        C, Ci, H, Hi = space_update(C, Ci, mean, m1, sched, sig[i + 1], x, xn - x)
        yield ("space", i, C, Ci, H, Hi, None, None)
        x, score = xn, (m1 - xn) / sig[i + 1] ** 2
