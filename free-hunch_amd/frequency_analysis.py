"""DCT-variance prior from a set of images (reference: do_frequency_analysis.py:1-72): the per-coefficient second
moment of the orthonormal 2-D DCT of images scaled to [-1, 1] - the `dct_variance.pt` that `CovarianceHessianBFGSDCT`
loads (online_update_bfgs.py:343).  The DCT runs on the gfx950 kernel (`fh_dct2d`), accumulation in float64."""
from __future__ import annotations

import torch

from . import _lib


def dct_variance(images_u8, device="cuda", batch=16):
    """images_u8: uint8 [N,3,S,S] (CPU or GPU).  Returns float32 [3,S,S] = mean over images of dct2(x)^2 with
    x = u8 / 127.5 - 1 (do_frequency_analysis.py:40-53 accumulates exactly this mean square)."""
    N, C, S, S2 = images_u8.shape
    assert C == 3 and S == S2
    dev = torch.device(device)
    ctx = _lib.Context.get(S, 3 * batch, 0, slot=1000)
    acc = torch.zeros(3, S, S, dtype=torch.float64, device=dev)
    for s in range(0, N, batch):
        x = images_u8[s: s + batch].to(dev).to(torch.float64) / 127.5 - 1
        n = x.shape[0]
        if n < batch:  # keep the plane count the context was built for
            x = torch.cat([x, torch.zeros(batch - n, 3, S, S, dtype=torch.float64, device=dev)])
        z = ctx.dct2d(x.contiguous())
        acc += (z[:n] ** 2).sum(0)
    return (acc / N).to(torch.float32)
