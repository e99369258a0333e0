"""Dense-matrix form of the Free Hunch covariance updates on the GPU (float64, batched (bs, d, d)).

Drop-in for the reference's dense helpers `update_covariance` / `update_bfgs`
(conditioning_utils/online_update_bfgs.py:377-463): same argument lists and return tuples, tensors on the GPU.
Every (bs,d,d) x (bs,d) product and every outer-product update runs in the hand-written HIP kernels
`fh_dense_matvec` / `fh_dense_rank2` (one streaming pass over the matrix each); the two dense inverses the reference
takes with `torch.linalg.inv` stay a library call (rocSOLVER through torch) - they are O(d^3) LAPACK work, not part of
the per-iteration cov-apply.  Differences in rounding only: the reference forms `(I - g dx de^T) C^-1 (I - g de dx^T)`
and `H' H^-1 score` with d^3 matrix products, here both are expanded into mat-vecs and one rank-2 update
(identical in exact arithmetic, ~1e-15 relative apart in float64).

No CPU fallback: tensors must live on the GPU and libfh_hip.so must load.
"""
from __future__ import annotations

import torch

from . import _lib

F64 = torch.float64


def _chk(A, name):
    if not (A.is_cuda and A.dtype == F64 and A.dim() == 3 and A.shape[1] == A.shape[2]):
        raise ValueError(f"{name}: expected a float64 GPU tensor of shape (bs, d, d), got {tuple(A.shape)} {A.dtype} "
                         f"on {A.device}")
    return A.contiguous()


def _vec(x, bs, d, name):
    x = torch.as_tensor(x)
    if x.shape != (bs, d):
        raise ValueError(f"{name}: expected shape ({bs}, {d}), got {tuple(x.shape)}")
    return x.to(F64).contiguous()


def matvec(A, x, trans=False, alpha=1.0, beta=0.0, out=None):
    """y[b] = alpha * op(A[b]) x[b] + beta * y[b]   (fh_dense_matvec).  A (bs,d,d), x (bs,d)."""
    lib = _lib.load()
    A = _chk(A, "A")
    bs, d = A.shape[0], A.shape[1]
    x = _vec(x, bs, d, "x")
    y = torch.empty_like(x) if out is None else out
    scratch = None
    if trans:
        scratch = torch.empty(lib.fh_dense_matvec_scratch_doubles(bs, d), dtype=F64, device=A.device)
    _lib.check(lib.fh_dense_matvec(_lib.ptr(A), _lib.ptr(x), _lib.ptr(y), _lib.ptr(scratch), bs, d, int(trans),
                                   float(alpha), float(beta), _lib.stream()), "fh_dense_matvec")
    return y


def rank2(A, u1=None, v1=None, a1=None, u2=None, v2=None, a2=None, scale=1.0, shift=0.0, out=None):
    """out[b] = scale * (A[b] + a1[b] u1[b] v1[b]^T + a2[b] u2[b] v2[b]^T) + shift * I   (fh_dense_rank2)."""
    lib = _lib.load()
    A = _chk(A, "A")
    bs, d = A.shape[0], A.shape[1]
    out = torch.empty_like(A) if out is None else out

    def term(u, v, a):
        if u is None:
            return None, None, None
        a = torch.as_tensor(a, dtype=F64, device=A.device).reshape(-1).expand(bs).contiguous()
        return _vec(u, bs, d, "u"), _vec(v, bs, d, "v"), a

    u1, v1, a1 = term(u1, v1, a1)
    u2, v2, a2 = term(u2, v2, a2)
    _lib.check(lib.fh_dense_rank2(_lib.ptr(A), _lib.ptr(out), bs, d, _lib.ptr(u1), _lib.ptr(v1), _lib.ptr(a1),
                                  _lib.ptr(u2), _lib.ptr(v2), _lib.ptr(a2), float(scale), float(shift), _lib.stream()),
               "fh_dense_rank2")
    return out


def _dot(a, b):
    return (a * b).sum(-1)


def update_covariance(samples, denoiser_cov, inv_denoiser_cov, hessian, inv_hessian, score_value, denoiser_mean,
                      schedule, t, tnext):
    """'Time' update of the dense forms (online_update_bfgs.py:377-410).  Returns
    (cov', cov'^-1, H', H'^-1, score', mean') with  cov'^-1 = cov^-1 + (s'^-2 - s^-2) I,  H'^-1 = H^-1 - (s'^2 - s^2) I,
    score' = H' H^-1 score,  mean' = x + s'^2 score'."""
    s, sn = float(schedule(t)), float(schedule(tnext))
    icov, ihess = _chk(inv_denoiser_cov, "inv_denoiser_cov"), _chk(inv_hessian, "inv_hessian")
    bs, d = icov.shape[0], icov.shape[1]
    new_icov = rank2(icov, shift=sn ** -2 - s ** -2)
    new_cov = torch.linalg.inv(new_icov)
    new_ihess = rank2(ihess, shift=-(sn ** 2 - s ** 2))
    new_hess = torch.linalg.inv(new_ihess)
    score = _vec(score_value, bs, d, "score_value")
    new_score = matvec(new_hess, matvec(ihess, score))
    new_mean = _vec(samples, bs, d, "samples") + sn ** 2 * new_score
    return new_cov, new_icov, new_hess, new_ihess, new_score, new_mean


def update_bfgs(denoiser_cov, inv_denoiser_cov, denoiser_mean_at_x, denoiser_mean_at_xnext, schedule, t, x, dx):
    """'Space' (BFGS) update of the dense forms (online_update_bfgs.py:412-463).  Returns (cov', cov'^-1, H', H'^-1):
    cov' = cov - (cov dx)(cov dx)^T / (dx^T cov dx) + g de de^T,  de = s^2 (m(x+dx) - m(x)),  g = 1 / (dx . de);
    cov'^-1 = (I - g dx de^T) cov^-1 (I - g de dx^T) + g dx dx^T;  H' = (cov'/s^2 - I)/s^2;  H'^-1 = inv(H' + 1e-10 I)."""
    s2 = float(schedule(t)) ** 2
    cov, icov = _chk(denoiser_cov, "denoiser_cov"), _chk(inv_denoiser_cov, "inv_denoiser_cov")
    bs, d = cov.shape[0], cov.shape[1]
    dx = _vec(dx, bs, d, "dx")
    de = s2 * (_vec(denoiser_mean_at_xnext, bs, d, "denoiser_mean_at_xnext")
               - _vec(denoiser_mean_at_x, bs, d, "denoiser_mean_at_x"))
    g = 1.0 / _dot(dx, de)
    cdx = matvec(cov, dx)          # cov dx
    dxc = matvec(cov, dx, trans=True)  # (dx^T cov)^T - the reference does not symmetrise cov
    new_cov = rank2(cov, cdx, dxc, -1.0 / _dot(dx, cdx), de, de, g)
    w_r = matvec(icov, de)              # cov^-1 de
    w_l = matvec(icov, de, trans=True)  # (de^T cov^-1)^T
    c = g * g * _dot(de, w_r) + g
    # cov^-1 - g dx w_l^T - g w_r dx^T + c dx dx^T  =  cov^-1 + dx (c dx - g w_l)^T + (-g) w_r dx^T
    new_icov = rank2(icov, dx, c[:, None] * dx - g[:, None] * w_l, 1.0, w_r, dx, -g)
    new_hess = rank2(new_cov, scale=1.0 / (s2 * s2), shift=-1.0 / s2)
    new_ihess = torch.linalg.inv(rank2(new_hess, shift=1e-10))
    return new_cov, new_icov, new_hess, new_ihess
