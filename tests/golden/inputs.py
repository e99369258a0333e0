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



def script(seed, shape, n_steps, sig0, neg_gamma_at=None):
    """A scripted alternation of time and space updates with seeded vectors (mimics the Heun call pattern)."""
    g = rng(seed)
    sig = np.geomspace(sig0, 0.5, n_steps + 1)
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


