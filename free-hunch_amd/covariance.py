"""Device-resident Free Hunch covariance state: the reference's ``CovarianceHessianBFGS`` /
``CovarianceHessianBFGSDCT`` (conditioning_utils/online_update_bfgs.py:7-374) re-designed for HBM.

Reference layout: four matrices (denoiser covariance C, C^-1, Hessian H, H^-1), each kept as
``diag + U U^T - V V^T`` with complex128 U, V (plain transposes), rebuilt on the CPU by two Woodbury
steps with ``scipy.linalg.sqrtm`` and re-uploaded after every update (8 factor matrices, 16 B/entry).

Here every representation is real:   X = diag(D) + diag(r) B M B^T diag(r)
  * B  [m_cap][d] float64, column-major, one *shared* base per family: C and C^-1 share ``Bc``,
    H and H^-1 share ``Bh``.  Columns are the raw BFGS pairs (de, C dx); they are written once and never
    rewritten - a Woodbury inverse only rescales rows (``r <- r / D``) and replaces the small matrix
    ``M <- -M (I + G M)^-1`` with ``G = B^T diag(r^2 / D) B``.
  * D, r [d] float64 per representation, M [m][m] float64 per representation (host master, device copy).
The complex square roots of the reference disappear: ``U U^T - V V^T`` is always real, only its
factorisation was complex.  2 bases instead of 8 factor matrices, 8 B/entry instead of 16 B, and no
factor ever crosses PCIe.  All d-sized work runs in libfh_hip.so (fh_rep_apply / fh_rep_invert /
fh_space_*); the m x m algebra (m <= 2 * number of space updates) stays on the host in float64.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib

F64 = torch.float64


class _Rep:
    """One representation over a shared base: diagonal D, row scale r, inner matrix M (m x m, device-resident in
    `M_dev[:m, :m]`; `.M` downloads it - tests and get_dense_matrices only)."""

    def __init__(self, D, m_cap):
        self.D = D
        self.r = torch.ones_like(D)
        self.m = 0
        self.M_dev = torch.zeros(m_cap, m_cap, dtype=F64, device=D.device)

    @property
    def M(self):
        return self.M_dev[: self.m, : self.m].cpu().numpy()

    def set_M(self, M):
        m = M.shape[0]
        self.m = m
        if m:
            self.M_dev[:m, :m].copy_(torch.from_numpy(np.ascontiguousarray(M)))

    def grow(self, m_cap):
        new = torch.zeros(m_cap, m_cap, dtype=F64, device=self.D.device)
        if self.m:
            new[: self.m, : self.m].copy_(self.M_dev[: self.m, : self.m])
        self.M_dev = new


class _Family:
    def __init__(self, d, m_cap, device):
        self.B = torch.empty(m_cap, d, dtype=F64, device=device)
        self.m = 0

    def grow(self, m_cap):
        new = torch.empty(m_cap, self.B.shape[1], dtype=F64, device=self.B.device)
        if self.m:
            new[: self.m].copy_(self.B[: self.m])
        self.B = new


def _csqrtm(A):
    """The reference's `sqrtm` wrapper (online_update_bfgs.py:67-71): scipy's principal square root, complex128."""
    if A.shape[0] == 0:
        return np.zeros_like(A, dtype=np.complex128)
    from scipy.linalg import sqrtm
    return np.asarray(sqrtm(A)).astype(np.complex128)


def _woodbury_factors(QU, QV, G):
    """The m x k algebra of `woodbury_inverse_from_diag_plus_lowrank_minus_lowrank` (online_update_bfgs.py:87-119) in the
    coordinates of the shared base.  The reference's factors are U = diag(r) B QU, V = diag(r) B QV (complex128, plain
    transposes); with G = B^T diag(r^2 / D) B the inverse's factors are diag(r / D) B QUi, diag(r / D) B QVi."""
    k = QU.shape[1]
    eye = np.eye(k)
    inner = np.linalg.inv(eye + QU.T @ G @ QU) if k else np.zeros((0, 0), np.complex128)
    inner = (inner + inner.T) / 2                              # :98-99
    QVi = QU @ _csqrtm(inner)                                  # :102-104
    K = QVi.T @ G @ QV                                         # :115  V_inv^T V
    inner2 = np.linalg.inv(eye - QV.T @ G @ QV + K.T @ K) if k else inner
    S2 = _csqrtm(inner2)                                       # :116-117
    QUi = QV @ S2 - QVi @ (K @ S2)                             # :118-119
    return QUi, QVi


def _woodbury_inner(M, G):
    """M_inv = -M (I + G M)^-1, symmetrised (the m x m part of online_update_bfgs.py:87-119)."""
    m = M.shape[0]
    if m == 0:
        return M
    # as k_woodbury_inner: I + G M is ill-conditioned on real trajectories (nearly dependent columns); float64 inverse, then
    # iterative refinement of X A = -M with extended-precision residuals
    LD = np.longdouble
    A = np.eye(m, dtype=LD) + G.astype(LD) @ M.astype(LD)
    Ainv = np.linalg.inv(A.astype(np.float64))
    out, prev = -M @ Ainv, np.inf
    for _ in range(12):
        R = (-M.astype(LD) - out.astype(LD) @ A).astype(np.float64)
        cur = float(np.abs(R).max())
        if not cur < prev:
            break
        out, prev = out + R @ Ainv, cur
    return 0.5 * (out + out.T)


class CovarianceHessianBFGS:
    """Same public surface as the reference class (online_update_bfgs.py:7-336); `device` is new.

    `init_denoiser_variance` is a scalar or a [data_dim] tensor; `init_noise_variance` a float (sigma_0^2).
    Vectors passed to the update methods have shape (1, 3, S, S) (batch 1, as the reference asserts)."""

    use_dct = False

    def __init__(self, init_denoiser_variance, init_noise_variance, data_dim, dtype=None, max_vector_count=None,
                 init_denoiser_cov_u=None, project_to_diagonal=False, use_precalculated_info=True, device=None,
                 m_cap=64, ctx_slot=0):
        assert init_denoiser_cov_u is None, "a non-empty initial factor is not supported"
        self.device = torch.device(device if device is not None else "cuda")
        S = int(round(math.sqrt(data_dim / 3)))
        if 3 * S * S != data_dim:
            raise ValueError(f"data_dim={data_dim} is not 3*S*S")
        self.S, self.data_dim = S, data_dim
        self.max_vector_count = max_vector_count
        self.project_to_diagonal = project_to_diagonal
        self.ctx_slot = ctx_slot
        self.ctx = _lib.Context.get(S, 3, _lib_max_cols(), ctx_slot)
        self.m_cap = m_cap
        d = data_dim
        var = torch.as_tensor(init_denoiser_variance, dtype=F64).reshape(-1).to(self.device)
        Dc = (torch.ones(d, dtype=F64, device=self.device) * var).contiguous()
        self.famC, self.famH = _Family(d, m_cap, self.device), _Family(d, m_cap, self.device)
        self.C, self.Ci = _Rep(Dc, m_cap), _Rep(torch.empty_like(Dc), m_cap)
        self.H, self.Hi = _Rep(torch.empty_like(Dc), m_cap), _Rep(torch.empty_like(Dc), m_cap)
        self._G = torch.zeros(m_cap, m_cap, dtype=F64, device=self.device)
        self._scal = torch.zeros(8, dtype=F64, device=self.device)
        self._t0, self._t1, self._t2 = (torch.empty(d, dtype=F64, device=self.device) for _ in range(3))
        self._tu0, self._tu1 = (torch.empty(d, dtype=F64, device=self.device) for _ in range(2))  # time-update work vectors
        self._st_cache = None
        # `0 < max_vector_count`: the reference keeps "the newest columns" of its sqrtm-mixed factors (:240-244), so those
        # factors must exist.  They are tracked in base coordinates (m x k complex matrices QU, QV per representation of
        # the C family - host algebra on a few dozen numbers, the d-sized work stays in the kernels); a count that the
        # column capacity can never reach needs no tracking and keeps the transfer-free fused path.
        self._track = max_vector_count is not None and 0 < max_vector_count < _lib_max_cols() // 2
        self._Q = {"C": (np.zeros((0, 0), np.complex128),) * 2, "Ci": (np.zeros((0, 0), np.complex128),) * 2}
        self._derive_from_cov(np.sqrt(init_noise_variance))

    # ------------------------------------------------------------------ helpers
    @property
    def k(self):
        """number of stored (u, v) pairs of the covariance (reference: vectors_denoiser_cov_u.shape[-1])"""
        return self._Q["C"][0].shape[1] if self._track else self.famC.m // 2

    def _ensure_capacity(self, need):
        if need <= self.m_cap:
            return
        cap = self.m_cap
        while cap < need:
            cap *= 2
        if cap > _lib_max_cols():
            hint = ""
            if self.max_vector_count is not None and 0 < self.max_vector_count < _lib_max_cols() // 2:
                hint = (f" (max_vector_count={self.max_vector_count} truncates the INNER matrix only: the base keeps every "
                        f"raw pair, so a run is limited to {_lib_max_cols() // 2} space updates; the reference has no such limit)")
            raise _lib.FhError(f"more than {_lib_max_cols()} factor columns requested" + hint)
        for fam in (self.famC, self.famH):
            fam.grow(cap)
        for rep in (self.C, self.Ci, self.H, self.Hi):
            rep.grow(cap)
        self._G = torch.zeros(cap, cap, dtype=F64, device=self.device)
        self.m_cap = cap
        self._st_cache = None

    def _invert(self, src, dst, fam, shift=0.0):
        """dst <- (src + shift*I)^-1 over the family's base; src.D is shifted in place.  No host round trip for
        m <= 64: the Gram matrix stays on the device and fh_woodbury_inner forms dst.M there."""
        m = fam.m
        assert src.m == m, (src.m, m)
        self.ctx.rep_invert(src.D, src.r, fam.B, shift, dst.D, dst.r, self._G, m)
        if m == 0:
            dst.m = 0
            return
        ctx = self.ctx
        rc = ctx.lib.fh_woodbury_inner(ctx.h, _lib.ptr(src.M_dev), src.M_dev.shape[1], _lib.ptr(self._G),
                                       self._G.shape[1], _lib.ptr(dst.M_dev), dst.M_dev.shape[1], m, _lib.stream())
        if rc == 0:
            dst.m = m
            return
        if rc != _lib.FH_ESIZE:
            _lib.check(rc, "fh_woodbury_inner")
        dst.set_M(_woodbury_inner(src.M, self._G[:m, :m].cpu().numpy()))  # m > 64: host LU

    def _invert_tracked(self, src, dst, src_key, dst_key, shift=0.0):
        """`_invert` for the C family with the reference's factor bookkeeping: the Gram matrix comes back to the host
        (m x m), the factor coordinates follow :87-119 and dst.M = QUi QUi^T - QVi QVi^T."""
        fam = self.famC
        m = fam.m
        self.ctx.rep_invert(src.D, src.r, fam.B, shift, dst.D, dst.r, self._G, m)
        if m == 0:
            dst.m = 0
            self._Q[dst_key] = self._Q[src_key]
            return
        G = self._G[:m, :m].cpu().numpy()
        QUi, QVi = _woodbury_factors(*self._Q[src_key], G)
        self._Q[dst_key] = (QUi, QVi)
        dst.set_M(self._real_inner(QUi @ QUi.T - QVi @ QVi.T))

    @staticmethod
    def _real_inner(M):
        """U U^T - V V^T is real whenever every stored pair is complete; after a truncation that splits a
        negative-curvature pair the reference carries a complex matrix internally and takes `.real` of each product
        (:202).  The kernels are real: keep the real part and say so."""
        im = float(np.abs(M.imag).max()) if M.size else 0.0
        if im > 1e-9 * max(1.0, float(np.abs(M.real).max()) if M.size else 0.0):
            import warnings
            warnings.warn("max_vector_count truncated a negative-curvature pair: the reference's inner matrix is complex "
                          f"here (|imag| = {im:.2e}); its real part is kept")
        M = np.ascontiguousarray(M.real)
        return 0.5 * (M + M.T)

    def _state(self):
        """the fh_cov_state of this object.  The pointer part is cached: the tensors behind it are replaced only when the
        column capacity grows (`_ensure_capacity` drops the cache); building it costs ~ 30 us of host time, and the
        lock-step sampler's update phase is bound by exactly that kind of per-image host work."""
        st = self._st_cache
        if st is None:
            st = _lib.FhCovState()
            st.d = self.data_dim
            st.ldm, st.ldg = self.C.M_dev.shape[1], self._G.shape[1]
            st.use_dct = int(self.use_dct)
            for i, rep in enumerate((self.C, self.Ci, self.H, self.Hi)):
                st.D[i], st.r[i], st.M[i] = rep.D.data_ptr(), rep.r.data_ptr(), rep.M_dev.data_ptr()
            st.Bc, st.Bh, st.G, st.scal = (self.famC.B.data_ptr(), self.famH.B.data_ptr(), self._G.data_ptr(),
                                           self._scal.data_ptr())
            st.t0, st.t1, st.t2 = self._t0.data_ptr(), self._t1.data_ptr(), self._t2.data_ptr()
            self._st_cache = st
        st.m_c, st.m_h = self.famC.m, self.famH.m
        st.project = int(bool(self.project_to_diagonal))
        return st

    def _apply(self, rep, fam, z, out):
        return self.ctx.rep_apply(rep.D, rep.r, fam.B, rep.M_dev, z, out, fam.m)

    def _derive_from_cov(self, sigma):
        """set_others_corresponding_to_current_denoiser_cov (:327-330) for an EMPTY factor."""
        assert self.famC.m == 0
        s2 = float(sigma) ** 2
        self.famH.m = 0
        for rep in (self.C, self.Ci, self.H, self.Hi):
            rep.r.fill_(1.0)
            rep.m = 0
        self._invert(self.C, self.Ci, self.famC)
        torch.div(self.C.D, s2, out=self.H.D)
        self.H.D.sub_(1.0).div_(s2)
        self._invert(self.H, self.Hi, self.famH)

    def _vec(self, x):
        return x.detach().to(device=self.device, dtype=F64).reshape(-1).contiguous()

    def _raw(self, x):
        """x itself when it already is a contiguous float64 tensor on this device (only its pointer is needed by the
        one-call updates), else the converted copy.  Four dispatcher calls per vector add up in the lock-step sampler."""
        if x.dtype is F64 and x.device == self.device and x.is_contiguous():
            return x
        return self._vec(x)

    def transform(self, x):
        return x

    def inverse_transform(self, x):
        return x

    def _fwd(self, v, out=None):
        return v

    def _bwd(self, v, out=None):
        return v

    # ------------------------------------------------------------------ :194-204
    def denoiser_cov_vector_dot(self, v, use_cuda=True):
        shape, dtype = v.shape, v.dtype
        z = self._fwd(self._vec(v))
        out = self._apply(self.C, self.famC, z, torch.empty_like(z))
        return self._bwd(out).reshape(shape).to(dtype)

    @staticmethod
    def denoiser_cov_vector_dot_batched(models, v, slot=0):
        """`denoiser_cov_vector_dot` of B covariance objects (same basis, same number of factor columns - a lock-step batch)
        on the B rows of v [B,3,S,S] in ONE kernel sequence: batched DCT (3B planes), fh_rep_apply_batched, batched IDCT.
        Same arithmetic per image as the per-object method (the batched kernels are bitwise equal to the single ones)."""
        m0 = models[0]
        B, S, d = len(models), m0.S, m0.data_dim
        assert v.shape[0] == B and all(mm.famC.m == m0.famC.m and mm.use_dct == m0.use_dct for mm in models)
        ctx = _lib.Context.get(S, 3 * B, 0, slot=7000 + 64 * slot + B)
        shape, dtype = v.shape, v.dtype
        z = v.detach().to(device=m0.device, dtype=F64).contiguous()
        if m0.use_dct:
            z = ctx.dct2d(z.view(3 * B, S, S))
        per = _lib.FhBatch()
        per.nimg = B
        for b, mm in enumerate(models):
            per.D[b], per.r[b], per.B[b], per.M[b] = (mm.C.D.data_ptr(), mm.C.r.data_ptr(), mm.famC.B.data_ptr(),
                                                      mm.C.M_dev.data_ptr())
        out = torch.empty_like(z)
        _lib.check(ctx.lib.fh_rep_apply_batched(ctx.h, C.byref(per), m0.C.M_dev.shape[1], z.data_ptr(), out.data_ptr(), d,
                                                m0.famC.m, _lib.stream()), "fh_rep_apply_batched")
        if m0.use_dct:
            out = ctx.dct2d(out.view(3 * B, S, S), inverse=True)
        return out.reshape(shape).to(dtype)

    # ------------------------------------------------------------------ :153-192
    def update_time_step(self, x_t, sigma_t, sigma_tnext, score_t, only_covariance=False):
        shape = x_t.shape
        assert shape[0] == 1, "Batch size must be 1"
        sigma_t, sigma_tnext = float(sigma_t), float(sigma_tnext)
        # the reference multiplies the increment by a float32 `torch.ones` (:166, :172): float32-rounded scalars
        shift_c = float(np.float32(sigma_tnext ** (-2) - sigma_t ** (-2)))
        shift_h = -float(np.float32(sigma_tnext ** 2 - sigma_t ** 2))
        if max(self.famC.m, self.famH.m) <= 64 and os.environ.get("FH_COV_STEPWISE") != "1" and not self._track:
            # one C call enqueues the whole update: the FORWARD-shift formulation (D / (1 + sD), M (I + sGM)^-1, double-double
            # refined; include/fh_hip.h).  The step-by-step path below (beyond 64 columns, FH_COV_STEPWISE=1, factor tracking
            # for 0 < max_vector_count) still follows the reference's route - shift the inverse representation, Woodbury back
            # with the 1/D-weighted Gram - whose float64 accuracy at d = 196608 with the DCT prior is that of the reference itself
            # (1e-6 .. 1e-4 below sigma = 0.2 on real states, profiles/r02_time_shift_accuracy.md), not the fused path's 1e-8
            ctx = self.ctx
            if only_covariance:
                _lib.check(ctx.lib.fh_cov_time_update(ctx.h, C.byref(self._state()), None, None, shift_c, shift_h, 0.0, 1,
                                                      None, None, None, None, _lib.stream()), "fh_cov_time_update")
                self.C.m = self.famC.m
                x = x_t.detach().to(device=self.device, dtype=F64)
                return x.clone(), x.clone()
            x, sc = self._raw(x_t), self._raw(score_t)  # locals keep converted copies alive until the launch
            wx, ws = self._tu0, self._tu1
            mean, score = (torch.empty(self.data_dim, dtype=F64, device=self.device) for _ in range(2))
            _lib.check(ctx.lib.fh_cov_time_update(ctx.h, C.byref(self._state()), _lib.ptr(x), _lib.ptr(sc), shift_c, shift_h,
                                                  sigma_tnext ** 2, 0, _lib.ptr(wx), _lib.ptr(ws), _lib.ptr(mean),
                                                  _lib.ptr(score), _lib.stream()), "fh_cov_time_update")
            self.C.m = self.famC.m
            self.H.m = self.famH.m
            return mean.reshape(shape), score.reshape(shape)
        if self._track:
            self._invert_tracked(self.Ci, self.C, "Ci", "C", shift=shift_c)
        else:
            self._invert(self.Ci, self.C, self.famC, shift=shift_c)
        if only_covariance:
            x = x_t.detach().to(device=self.device, dtype=F64)
            return x.clone(), x.clone()
        x = self._fwd(self._vec(x_t))
        s = self._fwd(self._vec(score_t))
        t = self._apply(self.Hi, self.famH, s, self._t0)  # old H^-1 score, before the diagonal moves
        self._invert(self.Hi, self.H, self.famH, shift=shift_h)
        new_score = self._apply(self.H, self.famH, t, torch.empty_like(t))
        new_mean = self.ctx.axpby(1.0, x, sigma_tnext ** 2, new_score, torch.empty_like(x))
        return self._bwd(new_mean).reshape(shape), self._bwd(new_score).reshape(shape)

    # ------------------------------------------------------------------ lock-step batches (one launch sequence for B images)
    @staticmethod
    def can_batch(models):
        """True when `update_*_step_batched` applies: one-call update path for every image (<= 62 columns, no factor
        tracking, no truncation), same basis / column counts / capacities - what a lock-step batch normally satisfies."""
        a = models[0]
        if os.environ.get("FH_COV_STEPWISE") == "1" or os.environ.get("FH_COV_BATCH", "1") == "0" or len(models) < 2:
            return False
        if max(a.famC.m, a.famH.m) + 2 > 64:
            return False
        if a.max_vector_count is not None and a.max_vector_count < 32:  # a truncation could trigger (k <= 31 on this path)
            return False
        return all((not mm._track) and mm.use_dct == a.use_dct and mm.famC.m == a.famC.m and mm.famH.m == a.famH.m
                   and mm.data_dim == a.data_dim and mm.m_cap == a.m_cap and mm.C.M_dev.shape[1] == a.C.M_dev.shape[1]
                   and bool(mm.project_to_diagonal) == bool(a.project_to_diagonal) and mm.device == a.device
                   and mm.max_vector_count == a.max_vector_count for mm in models)

    @staticmethod
    def _batch_ctx(models, slot):
        a = models[0]
        return _lib.Context.get(a.S, 3 * len(models), 64, slot=9000 + 64 * slot + len(models))

    @staticmethod
    def _batch_states(models):
        arr = (_lib.FhCovState * len(models))()
        for i, mm in enumerate(models):
            C.memmove(C.byref(arr[i]), C.byref(mm._state()), C.sizeof(_lib.FhCovState))
        return arr

    @staticmethod
    def update_time_step_batched(models, x_all, sigma_t, sigma_tnext, score_all, slot=0):
        """`update_time_step` of B covariance objects in ONE kernel sequence (fh_cov_time_update_batched; the kernels of the
        per-object call with the image as a grid dimension: bitwise the same result per image).  x_all, score_all:
        [B,3,S,S] float64 contiguous.  Returns (mean' [B,3,S,S], score' [B,3,S,S])."""
        a = models[0]
        B, d = len(models), a.data_dim
        sigma_t, sigma_tnext = float(sigma_t), float(sigma_tnext)
        shift_c = float(np.float32(sigma_tnext ** (-2) - sigma_t ** (-2)))  # float32-rounded like the reference (:166, :172)
        shift_h = -float(np.float32(sigma_tnext ** 2 - sigma_t ** 2))
        ctx = CovarianceHessianBFGS._batch_ctx(models, slot)
        x = x_all.detach().to(device=a.device, dtype=F64).contiguous()
        sc = score_all.detach().to(device=a.device, dtype=F64).contiguous()
        work = torch.empty(3, B, d, dtype=F64, device=a.device)
        mean, score = torch.empty_like(x), torch.empty_like(x)
        sts = CovarianceHessianBFGS._batch_states(models)
        _lib.check(ctx.lib.fh_cov_time_update_batched(ctx.h, B, sts, _lib.ptr(x), _lib.ptr(sc), shift_c, shift_h,
                                                      sigma_tnext ** 2, _lib.ptr(work), _lib.ptr(mean), _lib.ptr(score),
                                                      _lib.stream()), "fh_cov_time_update_batched")
        for mm in models:
            mm.C.m, mm.H.m = mm.famC.m, mm.famH.m
        return mean, score

    @staticmethod
    def update_space_step_batched(models, mean_x_all, mean_xn_all, sigma_t, x_all, xn_all, slot=0):
        """`update_space_step` of B covariance objects in ONE kernel sequence (fh_cov_space_update_batched)."""
        a = models[0]
        B, d = len(models), a.data_dim
        s2 = float(sigma_t) ** 2
        project = bool(a.project_to_diagonal)
        mc, mh = a.famC.m, a.famH.m
        for mm in models:
            mm._ensure_capacity(max(mc if project else mc + 2, mh + 2))
        ctx = CovarianceHessianBFGS._batch_ctx(models, slot)
        vecs = [t.detach().to(device=a.device, dtype=F64).contiguous() for t in (mean_x_all, mean_xn_all, x_all, xn_all)]
        work = torch.empty(3, B, d, dtype=F64, device=a.device)
        sts = CovarianceHessianBFGS._batch_states(models)
        _lib.check(ctx.lib.fh_cov_space_update_batched(ctx.h, B, sts, _lib.ptr(vecs[0]), _lib.ptr(vecs[1]), s2, _lib.ptr(vecs[2]),
                                                       _lib.ptr(vecs[3]), _lib.ptr(work), _lib.stream()),
                   "fh_cov_space_update_batched")
        for mm in models:
            if not project:
                mm.famC.m = mm.C.m = mm.Ci.m = mc + 2
            mm.famH.m = mm.H.m = mm.Hi.m = mh + 2

    # ------------------------------------------------------------------ :250-312
    def update_space_step(self, denoiser_mean_at_x, denoiser_mean_at_xnext, sigma_t, x, xnext):
        assert x.shape[0] == 1, "Batch size must be 1"
        ctx, lib = self.ctx, self.ctx.lib
        sigma_t = float(sigma_t)
        s2 = sigma_t ** 2
        d = self.data_dim
        if max(self.famC.m, self.famH.m) + 2 <= 64 and os.environ.get("FH_COV_STEPWISE") != "1" and not self._track:
            project = bool(self.project_to_diagonal)
            mc, mh = self.famC.m, self.famH.m
            self._ensure_capacity(max(mc if project else mc + 2, mh + 2))
            # locals keep the (possibly converted) inputs alive until the launch: a temporary freed between two
            # `_vec` calls would hand the same cached block to the next one
            mx, mxn, vx, vxn = (self._raw(t) for t in (denoiser_mean_at_x, denoiser_mean_at_xnext, x, xnext))
            _lib.check(lib.fh_cov_space_update(ctx.h, C.byref(self._state()), _lib.ptr(mx), _lib.ptr(mxn), s2,
                                               _lib.ptr(vx), _lib.ptr(vxn), _lib.stream()), "fh_cov_space_update")
            if not project:
                self.famC.m = self.C.m = self.Ci.m = mc + 2
            self.famH.m = self.H.m = self.Hi.m = mh + 2
            if self.max_vector_count is not None:
                self.drop_vectors(self.max_vector_count, sigma_t)
            return
        dx = self._fwd(ctx.axpby(1.0, self._vec(xnext), -1.0, self._vec(x), self._t0))
        dm = self._fwd(ctx.axpby(1.0, self._vec(denoiser_mean_at_xnext), -1.0, self._vec(denoiser_mean_at_x),
                                 self._t1))
        if dx.data_ptr() == self._t0.data_ptr():  # identity basis: _fwd returned the scratch itself
            dx, dm = dx.clone(), dm.clone()
        de = self._t1
        st = _lib.stream()
        _lib.check(lib.fh_space_prep(ctx.h, _lib.ptr(dm), s2, _lib.ptr(dx), _lib.ptr(de), _lib.ptr(self._scal), d, st),
                   "fh_space_prep")
        cdx = self._apply(self.C, self.famC, dx, self._t2)
        _lib.check(lib.fh_dot(ctx.h, _lib.ptr(cdx), _lib.ptr(dx), _lib.ptr(self._scal), 1, d, st), "fh_dot")
        # gamma = 1 / (dx.de) and q = dx.(C dx) stay on the device (self._scal): the commit kernel reads them there and
        # appends diag(gamma, -1/q) to the inner matrices itself
        project = bool(self.project_to_diagonal)
        mc, mh = self.famC.m, self.famH.m
        self._ensure_capacity(max(mc if project else mc + 2, mh + 2))
        Bc, Bh = self.famC.B, self.famH.B
        _lib.check(lib.fh_space_commit_dev(
            ctx.h, _lib.ptr(de), _lib.ptr(cdx), _lib.ptr(self._scal), s2, _lib.ptr(self.C.D), _lib.ptr(self.C.r),
            None if project else Bc[mc].data_ptr(), None if project else Bc[mc + 1].data_ptr(),
            _lib.ptr(self.H.D), _lib.ptr(self.H.r), Bh[mh].data_ptr(), Bh[mh + 1].data_ptr(),
            _lib.ptr(self.C.M_dev), self.C.M_dev.shape[1], mc, _lib.ptr(self.H.M_dev), self.H.M_dev.shape[1], mh,
            int(project), d, st), "fh_space_commit_dev")
        if not project:
            self.famC.m = self.C.m = mc + 2
        self.famH.m = self.H.m = mh + 2
        if self._track:
            if not project:
                # the new pair in base coordinates: u = sqrt(gamma) * column mc, v = column mc+1 / sqrt(q) (:271-272);
                # a negative gamma gives an imaginary u exactly as torch.sqrt does on the reference's complex scalars
                dxde, q = ctx.read_scalars(self._scal, 2)
                QU, QV = self._Q["C"]
                k = QU.shape[1]
                QU2, QV2 = np.zeros((mc + 2, k + 1), np.complex128), np.zeros((mc + 2, k + 1), np.complex128)
                QU2[:mc, :k], QV2[:mc, :k] = QU, QV
                QU2[mc, k] = np.sqrt(complex(1.0 / dxde))
                QV2[mc + 1, k] = 1.0 / np.sqrt(complex(q))
                self._Q["C"] = (QU2, QV2)
            self._invert_tracked(self.C, self.Ci, "C", "Ci")
        else:
            self._invert(self.C, self.Ci, self.famC)
        self._invert(self.H, self.Hi, self.famH)
        if self.max_vector_count is not None:
            self.drop_vectors(self.max_vector_count, sigma_t)

    # ------------------------------------------------------------------ :233-245
    def drop_vectors(self, max_vector_count, sigma):
        if max_vector_count == 0:
            self.famC.m = 0
            self._Q["C"] = (np.zeros((0, 0), np.complex128),) * 2
            self._derive_from_cov(sigma)
        elif self.k > max_vector_count:
            # keep the newest columns of U and V (:240-243).  The base keeps all its columns (they are raw pairs, never
            # rewritten); the truncation lives in the inner matrix M = QU QU^T - QV QV^T of rank <= 2 max_vector_count.
            assert self._track, "internal: truncation without factor tracking"
            n = max_vector_count
            QU, QV = self._Q["C"]
            QU, QV = QU[:, -n:], QV[:, -n:]
            self._Q["C"] = (QU, QV)
            self.C.set_M(self._real_inner(QU @ QU.T - QV @ QV.T))
            self._rederive_after_truncation(sigma)

    def _rederive_after_truncation(self, sigma):
        """set_others_corresponding_to_current_denoiser_cov (:327-330) for a non-empty factor: C^-1 by Woodbury, the
        Hessian re-built FROM the covariance (H = (C / s^2 - I) / s^2, i.e. the same factors / s^2), then H^-1."""
        s2 = float(sigma) ** 2
        m = self.famC.m
        self._invert_tracked(self.C, self.Ci, "C", "Ci")
        self._ensure_capacity(m)
        self.famH.B[:m].copy_(self.famC.B[:m])  # the Hessian family now spans the covariance's columns
        self.famH.m = m
        self.H.r.copy_(self.C.r)
        torch.div(self.C.D, s2, out=self.H.D)
        self.H.D.sub_(1.0).div_(s2)
        self.H.M_dev[:m, :m].copy_(self.C.M_dev[:m, :m] / (s2 * s2))
        self.H.m = m
        self._invert(self.H, self.Hi, self.famH)

    # ------------------------------------------------------------------ :320-325 (small d, tests only)
    def get_dense_matrices(self):
        out = []
        for rep, fam in ((self.C, self.famC), (self.Ci, self.famC), (self.H, self.famH), (self.Hi, self.famH)):
            W = (fam.B[: fam.m] * rep.r[None, :]).T
            M = rep.M_dev[: rep.m, : rep.m]
            out.append(torch.diag(rep.D) + W @ M @ W.T)
        return tuple(out)


_FILE_CACHE = {}


def _load_cached(path):
    """torch.load(weights_only=True) of a small data file, once per (path, modification time): the plugin is constructed per
    image (generate_conditional.py:120-128) and re-reading the prior variance 8 x per batch costs more than the guidance
    call it precedes."""
    key = (os.path.abspath(path), os.path.getmtime(path))
    if key not in _FILE_CACHE:
        _FILE_CACHE[key] = torch.load(path, weights_only=True)
    return _FILE_CACHE[key]


def _lib_max_cols():
    return 128  # column capacity of a context (Gram scratch grows with its square); Euler-100 needs 56


class CovarianceHessianBFGSDCT(CovarianceHessianBFGS):
    """All operations in the orthonormal 2-D DCT basis (online_update_bfgs.py:339-374); the prior diagonal is
    ``<data_dir>/dct_variance.pt`` when `use_precalculated_info`, else ones."""

    use_dct = True

    def __init__(self, data_dir, init_noise_variance, data_dim, dtype=None, max_vector_count=None, **kwargs):
        use_info = kwargs.pop("use_precalculated_info")
        if use_info:
            var = _load_cached(os.path.join(data_dir, "dct_variance.pt")).reshape(-1)
            if var.numel() != data_dim:
                raise ValueError(f"dct_variance.pt has {var.numel()} entries, data_dim is {data_dim}")
        else:
            var = torch.ones(data_dim)
        self.dct_variance = var
        super().__init__(var, init_noise_variance, data_dim, dtype, max_vector_count, **kwargs)

    def transform(self, x):
        return self.ctx.dct2d(x.detach().to(device=self.device, dtype=F64).contiguous()).to(x.dtype)

    def inverse_transform(self, x):
        return self.ctx.dct2d(x.detach().to(device=self.device, dtype=F64).contiguous(), inverse=True).to(x.dtype)

    def _fwd(self, v, out=None):
        return self.ctx.dct2d(v.view(3, self.S, self.S)).view(-1)

    def _bwd(self, v, out=None):
        return self.ctx.dct2d(v.view(3, self.S, self.S), inverse=True).view(-1)


class ScalarCovariance:
    """C = theta * I in image space - the covariance behind the reference's scalar-variance closed forms
    (`use_analytic_var_at_end`, conditioning_mechanisms.py:273-278).  Exposes just what the solver reads."""

    use_dct = False

    def __init__(self, theta, data_dim, device, ctx_slot=0):
        self.device = torch.device(device)
        self.data_dim = data_dim
        self.S = int(round(math.sqrt(data_dim / 3)))
        self.ctx = _lib.Context.get(self.S, 3, _lib_max_cols(), ctx_slot)
        self.famC = _Family.__new__(_Family)
        self.famC.m, self.famC.B = 0, torch.zeros(1, dtype=F64, device=self.device)
        self.C = _Rep(torch.full((data_dim,), float(theta), dtype=F64, device=self.device), 1)
