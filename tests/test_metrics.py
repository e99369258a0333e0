"""Metrics harness (generate_conditional.py:539-583): per-image PSNR / SSIM of uint8 images on the device
(fh_metrics_u8) and their gathered reduction.  scikit-image is absent from this image; the checker is a numpy /
scipy.ndimage restatement of the published SSIM algorithm with skimage's defaults (uniform 7x7 window, sample covariance,
K1 = 0.01, K2 = 0.03, border crop) - parity with skimage itself is unpinned."""
import numpy as np
import pytest
import scipy.ndimage
import torch

import inputs

pytestmark = pytest.mark.gpu


def _ssim_numpy(x, y, win=7, k1=0.01, k2=0.03, data_range=255.0):
    vals = []
    for c in range(x.shape[0]):
        a, b = x[c].astype(np.float64), y[c].astype(np.float64)
        f = lambda t: scipy.ndimage.uniform_filter(t, size=win, mode="reflect")
        ux, uy = f(a), f(b)
        n_p = win * win
        cn = n_p / (n_p - 1)
        vx, vy, vxy = cn * (f(a * a) - ux * ux), cn * (f(b * b) - uy * uy), cn * (f(a * b) - ux * uy)
        c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        pad = (win - 1) // 2
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def _psnr_numpy(x, y):
    mse = np.mean((x.astype(np.float64) - y.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize("shape", [(3, 3, 48, 40), (2, 3, 256, 256), (1, 1, 7, 9), (5, 3, 33, 130)])
def test_psnr_and_ssim_kernel_matches_the_published_definitions(shape):
    """fh_metrics_u8 against the restatement: ragged sizes (tiles that stick out, a single window row), 256 x 256 x 3."""
    from free_hunch_amd.pipeline import metrics_u8, psnr_u8, ssim_u8
    dev = torch.device("cuda:0")
    g = inputs.rng(5 + sum(shape))
    a = (torch.rand(*shape, generator=g) * 255).to(torch.uint8)
    smooth = torch.nn.functional.avg_pool2d(a.float(), 5, stride=1, padding=2)
    b = (smooth + 12 * torch.randn(smooth.shape, generator=g)).clamp(0, 255).to(torch.uint8)
    ps, ss = metrics_u8(a.to(dev), b.to(dev))
    assert ps.dtype == torch.float64 and ss.dtype == torch.float64
    for i in range(a.shape[0]):
        assert abs(float(ps[i]) - _psnr_numpy(a[i].numpy(), b[i].numpy())) < 1e-9
        assert abs(float(ss[i]) - _ssim_numpy(a[i].numpy(), b[i].numpy())) < 1e-10
    assert float(ssim_u8(a.to(dev), a.to(dev)).min()) > 1 - 1e-12  # identical images
    assert float(psnr_u8(a.to(dev), a.to(dev)).min()) > 100.0         # mse clamps at 1e-12
    with pytest.raises(Exception):
        metrics_u8(a, b)  # CPU tensors: no fallback
