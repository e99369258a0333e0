"""Measurement operators of the Free Hunch path on the device (reference:
measurement_utils/measurements.py:25-246, utils_sisr.py:9-96, resizer.py:8-160).

Same registry (`get_operator(name, **kwargs)`), same constructor keywords, same attributes the solvers read
(`name`, `sigma_s`, `in_shape`, `mask`, `scale_factor`).  The reference applies the blur through FFTs of a
zero-padded, rolled PSF (`p2o`); a circular convolution with the PSF's non-zero taps is the same linear map,
so the operators carry a sparse tap list (`taps`) that libfh_hip.so consumes.  `pre_calculated`
(FB, FBC, F2B, FBFy) is still offered for code that wants the transfer functions, computed lazily.
"""
from __future__ import annotations

import os
from functools import partial

import numpy as np
import scipy.io
import torch

from . import _lib

F64 = torch.float64
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
KERNEL_DIR = os.path.join(_DATA, "kernels")

__OPERATOR__ = {}


def register_operator(name):
    def wrapper(cls):
        if __OPERATOR__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        cls.name = name
        __OPERATOR__[name] = cls
        return cls
    return wrapper


def get_operator(name, **kwargs):
    if __OPERATOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __OPERATOR__[name](**kwargs)


class _TapList:
    def __init__(self, k, device):
        ys, xs = np.nonzero(k)
        cy, cx = k.shape[0] // 2, k.shape[1] // 2
        self.n = int(ys.size)
        hy, hx = int(np.abs(ys - cy).max()), int(np.abs(xs - cx).max())
        # fh_conv_circ halo code: for 1-D lists -(h+1) (column kernel, dx = 0) / -(h+101) (row kernel), else both extents
        # (1000 + 64 hy + hx), so that the kernel stages only the halo the taps reach
        self.halo = -(hy + 1) if (hx == 0 and hy > 0) else (-(hx + 101) if (hy == 0 and hx > 0) else 1000 + 64 * hy + hx)
        self.dy = torch.from_numpy((ys - cy).astype(np.int32)).to(device)
        self.dx = torch.from_numpy((xs - cx).astype(np.int32)).to(device)
        self.w = torch.from_numpy(np.ascontiguousarray(k[ys, xs], dtype=np.float64)).to(device)


class Taps(_TapList):
    """Non-zero PSF entries as (dy, dx, w) with the PSF centre at index size//2 (p2o's roll, utils_sisr.py:37-38).
    A numerically rank-1 PSF (the shipped Gaussian: second singular value 1e-8 of the first once rounded to float32,
    1e-16 before) is additionally split into a column and a row tap list (`sep`), so that the blur runs as two 1-D
    passes: 2 x 25 instead of 625 taps per pixel.  The rank-1 factor differs from the float32 taps by < 1e-7
    relative, the same size as the float32 rounding of the PSF / the complex64 OTF the reference blurs with."""

    def __init__(self, kernel, device):
        k = np.asarray(kernel, dtype=np.float32).astype(np.float64)  # the reference holds the PSF in float32
        super().__init__(k, device)
        self.kernel = torch.from_numpy(k.astype(np.float32))
        self.sep = None
        if min(k.shape) > 1:
            u, sv, vt = np.linalg.svd(k)
            if sv[1] <= 1e-6 * sv[0]:
                col, row = u[:, 0] * np.sqrt(sv[0]), vt[0] * np.sqrt(sv[0])
                if col.sum() < 0:
                    col, row = -col, -row
                col[np.abs(col) < 1e-12 * np.abs(col).max()] = 0.0
                row[np.abs(row) < 1e-12 * np.abs(row).max()] = 0.0
                self.sep = (_TapList(col[:, None], device), _TapList(row[None, :], device))


def dct_basis_longdouble(S):
    """Orthonormal DCT-II matrix C[k][n] = s_k cos(pi (2n + 1) k / 2S) in extended precision (the same formula the
    context's device basis is built from, csrc/fh_kernels.hip: fh_context_create)."""
    ld = np.longdouble
    k = np.arange(S, dtype=ld)[:, None]
    n = np.arange(S, dtype=ld)[None, :]
    C = np.cos(np.arccos(ld(-1)) * (2 * n + 1) * k / (2 * ld(S)))  # arccos(-1) = pi in extended precision
    C *= np.sqrt(ld(2) / ld(S))
    C[0] *= np.sqrt(ld(0.5))
    return C


def folded_dct_blur_bases(col_dy, col_w, row_dx, row_w, S):
    """The two 1-D passes of a separable circular blur folded into the 2-D DCT that follows / precedes them inside the CG
    operator A C A^T (C lives in the DCT basis):  with the blur A(X) = F_col X F_row^T,
        dct2(A^T u)  = P_col u P_row^T,        A(idct2(v)) = P_col^T v P_row,        P = C_dct F^T,
    so the four image passes of the two blurs disappear into the (dense) DCT passes.  F is circulant,
    F[i][i'] = sum_t w_t [i' = (i - d_t) mod S], hence P[k][n] = sum_t w_t C[k][(n - d_t) mod S] - formed in extended
    precision and rounded once.  Returns float64 (P_row, P_col, P_row^T, P_col^T), C-contiguous [S][S]."""
    C = dct_basis_longdouble(S)

    def fold(offsets, weights):
        P = np.zeros((S, S), dtype=np.longdouble)
        for d, w in zip(offsets, weights):
            P += np.longdouble(w) * np.roll(C, int(d), axis=1)
        return P

    P_row, P_col = fold(row_dx, row_w), fold(col_dy, col_w)
    f = lambda M: np.ascontiguousarray(M.astype(np.float64))
    return f(P_row), f(P_col), f(P_row.T), f(P_col.T)


def pack_symmetric_halves(P):
    """If the S x S basis P has the DCT's mirror symmetry P[k][S-1-n] = (-1)^k P[k][n] (the DCT itself; a SYMMETRIC blur
    folded into it), return the packed half bases of fh_problem.fold_sym = 1 (include/fh_hip.h): forward [Pe; Po] with
    Pe[j][n] = P[2j][n], Po[j][n] = P[2j+1][n], and inverse [Qe; Qo] = [Pe^T; Po^T], each float64 [2][S/2][S/2]; else None."""
    S = P.shape[0]
    if S % 128 != 0:  # k_dct_sym works on K chunks of 64 of the half range
        return None
    H = S // 2
    sign = np.where(np.arange(S) % 2 == 0, 1.0, -1.0)[:, None]
    if np.abs(P[:, ::-1] * sign - P).max() > 1e-13 * np.abs(P).max():
        return None
    Pe, Po = P[0::2, :H], P[1::2, :H]
    fwd = np.ascontiguousarray(np.stack([Pe, Po]))
    inv = np.ascontiguousarray(np.stack([Pe.T, Po.T]))
    return fwd, inv


_FOLD_CACHE = {}


class LinearOperator:
    device = None

    ctx_slot = 0  # scratch-context slot; the batched sampler gives every concurrent image its own

    def _ctx(self):
        S = self.in_shape[-1]
        return _lib.Context.get(S, 3, 128, self.ctx_slot)

    def _conv(self, x, stride=1, adjoint=False):
        """float64 circular convolution of an NCHW tensor (N = 1) with self.taps; returns float64."""
        S = self.in_shape[-1]
        x64 = x.detach().to(device=self.device, dtype=F64).contiguous()
        so = S if (adjoint or stride == 1) else S // stride
        out = torch.empty(x64.shape[0], x64.shape[1], so, so, dtype=F64, device=self.device)
        planes = x64.shape[0] * x64.shape[1]
        if self.taps.sep is not None and stride == 1:
            first, second = self.taps.sep if not adjoint else self.taps.sep[::-1]
            tmp = torch.empty_like(out)
            self._ctx().conv(x64, tmp, first, planes, 1, adjoint)
            self._ctx().conv(tmp, out, second, planes, 1, adjoint)
        else:
            self._ctx().conv(x64, out, self.taps, planes, stride, adjoint)
        return out

    def folded_bases(self):
        """(device copies of `folded_dct_blur_bases` for this operator's separable PSF, packed-symmetric flag) - None when
        the PSF is not rank-1 or the operator decimates; built once per (PSF, size, device), shared by all instances."""
        sep = getattr(getattr(self, "taps", None), "sep", None)
        if sep is None or self.name == "super_resolution" or os.environ.get("FH_NO_FOLD") == "1":
            return None
        S = self.in_shape[-1]
        col, row = sep
        key = (S, str(self.device), col.dy.cpu().numpy().tobytes(), col.w.cpu().numpy().tobytes(),
               row.dx.cpu().numpy().tobytes(), row.w.cpu().numpy().tobytes())
        if key not in _FOLD_CACHE:
            mats = folded_dct_blur_bases(col.dy.cpu().numpy(), col.w.cpu().numpy(), row.dx.cpu().numpy(),
                                         row.w.cpu().numpy(), S)
            hw, hh = pack_symmetric_halves(mats[0]), pack_symmetric_halves(mats[1])
            sym = hw is not None and hh is not None and os.environ.get("FH_DCT_NOSYM") is None
            if sym:  # (fold_fwd_w, fold_fwd_h, fold_inv_w, fold_inv_h) as packed halves
                mats = (hw[0], hh[0], hw[1], hh[1])
            _FOLD_CACHE[key] = (tuple(torch.from_numpy(m).to(self.device) for m in mats), sym)
        return _FOLD_CACHE[key]

    def _noise(self, y, noiseless):
        if not noiseless:
            y = y + self.sigma_s.to(y.dtype) * torch.randn_like(y)
        return y

    @property
    def pre_calculated(self):
        """(FB, FBC, F2B, FBFy) as utils_sisr.pre_calculate would return them for the last measurement."""
        k = self.taps.kernel.to(self.device)
        S = self.in_shape[-1]
        otf = torch.zeros(1, 1, S, S, device=self.device)
        otf[..., : k.shape[0], : k.shape[1]] = k
        otf = torch.roll(otf, (-(k.shape[0] // 2), -(k.shape[1] // 2)), dims=(-2, -1))
        FB = torch.fft.fftn(otf, dim=(-2, -1))
        FBC = torch.conj(FB)
        y = getattr(self, "_last_y", None)
        FBFy = None
        if y is not None:
            sf = getattr(self, "scale_factor", 1) if self.name == "super_resolution" else 1
            up = torch.zeros(y.shape[0], y.shape[1], S, S, device=self.device, dtype=y.dtype)
            up[..., ::sf, ::sf] = y
            FBFy = FBC * torch.fft.fftn(up, dim=(-2, -1))
        return FB, FBC, torch.abs(FB) ** 2, FBFy


class _BlurOperator(LinearOperator):
    kernel_file = None

    def __init__(self, in_shape, kernel_size, intensity, sigma_s, device, **kwargs):
        self.device = torch.device(device)
        self.kernel_size = kernel_size
        self.kernel = np.load(os.path.join(KERNEL_DIR, self.kernel_file))
        self.taps = Taps(self.kernel, self.device)
        self.sigma_s = torch.Tensor([sigma_s]).to(self.device)
        self.in_shape = in_shape

    def forward(self, data, flatten=False, noiseless=False):
        y = self._noise(self._conv(data).to(data.dtype), noiseless)
        self._last_y = y
        if flatten:
            return y, y.reshape(y.shape[0], -1)
        return y

    def transpose(self, y, flatten=False):
        if flatten:
            y = y.reshape(y.shape[0], *self.in_shape[-3:])
        return self._conv(y, adjoint=True).to(y.dtype)

    def forward_adjoint(self, v):
        """exact adjoint of the noiseless `forward`, for DPS"""
        return self._conv(v, adjoint=True).to(v.dtype)

    def get_kernel(self):
        return self.taps.kernel.view(1, 1, self.kernel_size, self.kernel_size).to(self.device)


@register_operator(name="gaussian_blur")
class GaussialBlurOperator(_BlurOperator):  # (sic) the reference's class name, measurements.py:164
    kernel_file = "gaussian_ks61_std3.0.npy"


@register_operator(name="motion_blur")
class MotionBlurOperator(_BlurOperator):  # measurements.py:126 (weights come from the shipped .npy, :135-136)
    kernel_file = "motion_ks61_std0.5.npy"


def _cubic(x):
    ax = np.abs(x)
    return ((1.5 * ax ** 3 - 2.5 * ax ** 2 + 1) * (ax <= 1)
            + (-0.5 * ax ** 3 + 2.5 * ax ** 2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2)))


def resize_matrix(n_in, scale):
    """[n_out, n_in] float32 matrix of the antialiased cubic Resizer along one axis (resizer.py:103-160)."""
    n_out = int(np.ceil(n_in * scale))
    width = 4.0 / scale
    centre = (np.arange(1, n_out + 1) - (n_out - n_in * scale) / 2) / scale + 0.5 * (1 - 1 / scale)
    first = np.floor(centre - width / 2)
    pos = (first[:, None] + np.arange(np.ceil(width) + 2) - 1).astype(np.int64)
    wgt = scale * _cubic(scale * (centre[:, None] - pos - 1))
    norm = wgt.sum(1, keepdims=True)
    norm[norm == 0] = 1.0
    wgt = (wgt / norm).astype(np.float32)
    fold = np.concatenate((np.arange(n_in), np.arange(n_in - 1, -1, -1)))
    pos = fold[np.mod(pos, 2 * n_in)]
    R = np.zeros((n_out, n_in), dtype=np.float32)
    np.add.at(R, (np.repeat(np.arange(n_out), pos.shape[1]), pos.reshape(-1)), wgt.reshape(-1))
    return torch.from_numpy(R)


@register_operator(name="super_resolution")
class SuperResolutionOperator(LinearOperator):  # measurements.py:87-123
    def __init__(self, in_shape, scale_factor, sigma_s, device, **kwargs):
        self.device = torch.device(device)
        self.scale_factor = int(scale_factor)
        self.sigma_s = torch.Tensor([sigma_s]).to(self.device)
        kernels = scipy.io.loadmat(os.path.join(KERNEL_DIR, "kernels_bicubicx234.mat"))["kernels"]
        k_index = self.scale_factor - 2 if self.scale_factor < 5 else 2
        self.kernel = kernels[0, k_index].astype(np.float64)
        self.taps = Taps(self.kernel, self.device)
        self.in_shape = in_shape
        self.out_shape = (1, 3, int(in_shape[-2] / self.scale_factor), int(in_shape[-1] / self.scale_factor))
        self._R = resize_matrix(in_shape[-1], 1.0 / self.scale_factor).to(self.device)

    def forward(self, data, flatten=False, noiseless=False):
        # the *measurement* is the antialiased bicubic Resizer (one-off per image, outside the solver loop);
        # the solver's A is blur(25x25 bicubic PSF) + decimation
        R = self._R.to(data.dtype)
        y = torch.einsum("oh,nchw->ncow", R, data.to(self.device))
        y = torch.einsum("pw,ncow->ncop", R, y)
        y = self._noise(y, noiseless)
        self._last_y = y
        if flatten:
            return y, y.reshape(y.shape[0], -1)
        return y

    def transpose(self, y, flatten=False):
        if flatten:
            y = y.reshape(y.shape[0], *self.out_shape[-3:])
        return self._conv(y, stride=self.scale_factor, adjoint=True).to(y.dtype)

    def forward_adjoint(self, v):
        """exact adjoint of the noiseless `forward` (the antialiased bicubic Resizer), for DPS"""
        R = self._R.to(v.dtype)
        return torch.einsum("oh,pw,ncop->nchw", R, R, v.to(self.device))

    def get_kernel(self):
        return self.taps.kernel.view(1, 1, *self.taps.kernel.shape).to(self.device)


@register_operator(name="inpainting")
class InpaintingOperator(LinearOperator):  # measurements.py:204-246
    def __init__(self, device, sigma_s, mask_opt, mask=None, **kwargs):
        self.device = torch.device(device)
        self.sigma_s = torch.Tensor([sigma_s]).to(self.device)
        self.in_shape = (1, 3, mask_opt["image_size"], mask_opt["image_size"])
        self.mask = self.generate_mask(mask_opt) if mask is None else mask.to(self.device)

    def forward(self, data, flatten=False, noiseless=False):
        y = self._noise(data.clone(), noiseless) * self.mask.to(data.dtype)
        if flatten:
            idx = torch.where(self.mask > 0)
            return y, y[..., idx[-3], idx[-2], idx[-1]]
        return y

    def forward_adjoint(self, v):
        return v * self.mask.to(v.dtype)

    def transpose(self, data, flatten=False):
        y = data.clone()
        if flatten:
            idx = torch.where(self.mask > 0)
            x = torch.zeros(y.shape[0], *self.in_shape[-3:], device=self.device)
            x[..., idx[-3], idx[-2], idx[-1]] = y
            return x
        return y

    def generate_mask(self, mask_opt):
        img = torch.randn(*self.in_shape).to(self.device)  # consumes the global torch RNG like the reference (:244)
        return MaskGenerator(**mask_opt)(img)


class MaskGenerator:  # measurements.py:248-318 (random / box / extreme; global numpy RNG, as the reference)
    def __init__(self, mask_type, mask_len_range=None, mask_prob_range=None, image_size=256, margin=(16, 16)):
        assert mask_type in ["box", "random", "both", "extreme"]
        self.mask_type, self.mask_len_range, self.mask_prob_range = mask_type, mask_len_range, mask_prob_range
        self.image_size, self.margin = image_size, margin

    def __call__(self, img):
        if self.mask_type == "random":
            return self._random(img)
        mask = self._box(img)
        return 1.0 - mask if self.mask_type == "extreme" else mask

    def _random(self, img):
        n = self.image_size ** 2
        prob = np.random.uniform(*self.mask_prob_range)
        keep = torch.ones(n)
        keep[np.random.choice(n, int(n * prob), replace=False)] = 0
        return keep.view(1, 1, self.image_size, self.image_size).expand(img.shape[0], img.shape[1], -1, -1) \
                   .contiguous().to(img.device)

    def _box(self, img):
        lo, hi = int(self.mask_len_range[0]), int(self.mask_len_range[1])
        h, w = np.random.randint(lo, hi), np.random.randint(lo, hi)
        mh, mw = self.margin
        t = np.random.randint(mh, self.image_size - mh - h)
        l = np.random.randint(mw, self.image_size - mw - w)
        mask = torch.ones_like(img)
        mask[..., t:t + h, l:l + w] = 0
        return mask
