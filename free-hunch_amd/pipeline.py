"""Per-rank image driver (reference: generate_conditional.py:289-414, 499-593) re-organised for sharded, barrier-free
operation: image i -> rank i mod world; lock-step batches per rank; ONE all_gather of the uint8 outputs at the end."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(total, rank, world):
    """Static shard i = rank (mod world).  No padding: ranks may differ by one image."""
    return list(range(rank, total, world))


def gather_images(local_u8, local_idx, total, device):
    """One all_gather of uint8 [n_local,3,S,S] (+ int64 indices); returns the [total,3,S,S] tensor ordered by index on
    every rank.  Ranks with fewer images pad with index -1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = torch.zeros((total,) + tuple(local_u8.shape[1:]), dtype=torch.uint8, device=local_u8.device)
        out[torch.as_tensor(local_idx, dtype=torch.long, device=local_u8.device)] = local_u8
        return out
    world = dist.get_world_size()
    n_max = (total + world - 1) // world
    shape = tuple(local_u8.shape[1:])
    buf = torch.zeros((n_max,) + shape, dtype=torch.uint8, device=device)
    idx = torch.full((n_max,), -1, dtype=torch.int64, device=device)
    n = local_u8.shape[0]
    buf[:n] = local_u8.to(device)
    idx[:n] = torch.as_tensor(local_idx, dtype=torch.int64, device=device)
    all_buf = [torch.empty_like(buf) for _ in range(world)]
    all_idx = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(all_buf, buf)
    dist.all_gather(all_idx, idx)
    out = torch.zeros((total,) + shape, dtype=torch.uint8, device=device)
    for b, i in zip(all_buf, all_idx):
        keep = i >= 0
        out[i[keep]] = b[keep]
    return out


def psnr_u8(a, b):
    mse = ((a.float() - b.float()) ** 2).mean(dim=(1, 2, 3)).clamp_min(1e-12)
    return 10 * torch.log10(255.0 ** 2 / mse)


def ssim_u8(a, b, win_size=7, k1=0.01, k2=0.03):
    """Mean structural similarity of uint8 image batches [N, C, H, W], the published algorithm with the defaults of
    `skimage.metrics.structural_similarity(x, y, data_range=255, channel_axis=0)` as the reference calls it
    (generate_conditional.py:546): 7x7 uniform window, sample covariance (NP / (NP - 1)), K1 = 0.01, K2 = 0.03,
    float64, the (win_size - 1) / 2 border cropped, mean over pixels then over channels.  scikit-image is not
    installed in this image, so this is pinned to a scipy.ndimage restatement (tests/test_metrics.py), not to skimage
    itself: parity unpinned at that boundary."""
    x, y = a.to(torch.float64), b.to(torch.float64)
    n_p = win_size * win_size
    cov_norm = n_p / (n_p - 1.0)
    pool = lambda t: torch.nn.functional.avg_pool2d(t, win_size, stride=1)  # 'valid' window means = the cropped region
    ux, uy = pool(x), pool(y)
    vx = cov_norm * (pool(x * x) - ux * ux)
    vy = cov_norm * (pool(y * y) - uy * uy)
    vxy = cov_norm * (pool(x * y) - ux * uy)
    c1, c2 = (k1 * 255.0) ** 2, (k2 * 255.0) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return s.mean(dim=(2, 3)).mean(dim=1)


def list_images(path):
    exts = (".png", ".jpg", ".jpeg")
    files = []
    for root, _dirs, names in os.walk(path):
        files += [os.path.join(root, n) for n in names if n.lower().endswith(exts)]
    return sorted(files)


def load_image_u8(path, size):
    import PIL.Image
    img = PIL.Image.open(path).convert("RGB")
    if img.size != (size, size):
        img = img.resize((size, size), PIL.Image.BICUBIC)
    return torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1)
