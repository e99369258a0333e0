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


def metrics_u8(a, b):
    """(PSNR [N], SSIM [N]) float64 of uint8 image batches [N, C, H, W] on the device - the reference's per-image metrics
    (generate_conditional.py:543-551; skimage.metrics.structural_similarity(x, y, data_range=255, channel_axis=0): 7 x 7 uniform
    window, sample covariance, K1 = 0.01, K2 = 0.03, windows inside the image, mean over positions then channels), one
    `fh_metrics_u8` call.  LPIPS needs a network download and is not offered.  scikit-image is absent from this image, so
    the kernel is pinned to a scipy.ndimage restatement of the published algorithm (tests/test_metrics.py), not to skimage
    itself: parity unpinned at that boundary."""
    from . import _lib
    if not (a.is_cuda and b.is_cuda):
        raise _lib.FhError("metrics_u8 runs on the device (libfh_hip.so); there is no CPU fallback")
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and a.shape == b.shape and a.dim() == 4
    lib = _lib.load()
    a, b = a.contiguous(), b.contiguous()
    N, C, H, W = a.shape
    scratch = torch.empty(int(lib.fh_metrics_scratch_doubles(N, C, H, W)), dtype=torch.float64, device=a.device)
    ssim = torch.empty(N, dtype=torch.float64, device=a.device)
    psnr = torch.empty(N, dtype=torch.float64, device=a.device)
    _lib.check(lib.fh_metrics_u8(a.data_ptr(), b.data_ptr(), N, C, H, W, scratch.data_ptr(), ssim.data_ptr(), psnr.data_ptr(),
                                 _lib.stream()), "fh_metrics_u8")
    return psnr, ssim


def psnr_u8(a, b):
    return metrics_u8(a, b)[0]


def ssim_u8(a, b):
    return metrics_u8(a, b)[1]


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
