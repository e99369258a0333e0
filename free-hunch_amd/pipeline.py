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


def gather_images(local_u8, local_idx, total, device, partial_sums=None):
    """THE exchange of a run: ONE all_gather (RCCL over xGMI on the GPUs, gloo in the CPU rehearsals) of one uint8 payload
    per rank = [partial sums as float64 | per image: int64 global index, uint8 pixels].  Returns the [total,3,S,S] tensor
    ordered by index on every rank - and, when `partial_sums` (a 1-D float64 tensor, e.g. the metric sums of
    generate_conditional.py:557-569) is given, also their sum over ranks, so that no separate all_reduce is needed.
    Ranks with fewer images pad with index -1."""
    shape = tuple(local_u8.shape[1:])
    nsum = 0 if partial_sums is None else int(partial_sums.numel())
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = torch.zeros((total,) + shape, dtype=torch.uint8, device=local_u8.device)
        out[torch.as_tensor(local_idx, dtype=torch.long, device=local_u8.device)] = local_u8
        return out if partial_sums is None else (out, partial_sums.clone())
    world = dist.get_world_size()
    n_max = (total + world - 1) // world
    img_bytes = int(np.prod(shape))
    rec = 8 + img_bytes                       # int64 index + pixels
    payload = torch.zeros(8 * nsum + n_max * rec, dtype=torch.uint8, device=device)
    if nsum:
        payload[: 8 * nsum] = partial_sums.detach().to(device=device, dtype=torch.float64).contiguous().view(torch.uint8)
    body = payload[8 * nsum:].view(n_max, rec)
    idx = torch.full((n_max,), -1, dtype=torch.int64, device=device)
    n = local_u8.shape[0]
    idx[:n] = torch.as_tensor(local_idx, dtype=torch.int64, device=device)
    body[:, :8] = idx.view(torch.uint8).view(n_max, 8)
    body[:n, 8:] = local_u8.to(device).reshape(n, img_bytes)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)
    out = torch.zeros((total,) + shape, dtype=torch.uint8, device=device)
    sums = torch.zeros(nsum, dtype=torch.float64, device=device)
    for g in gathered:
        if nsum:
            sums += g[: 8 * nsum].clone().view(torch.float64)
        gb = g[8 * nsum:].view(n_max, rec)
        i = gb[:, :8].contiguous().view(torch.int64).view(n_max)
        keep = i >= 0
        out[i[keep]] = gb[keep][:, 8:].reshape((-1,) + shape)
    return out if partial_sums is None else (out, sums)


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
