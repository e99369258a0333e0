"""ctypes binding of libfh_hip.so (include/fh_hip.h).  There is no CPU fallback: if the library is missing or a
call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfh_hip.so")

c_dp = C.c_void_p  # device pointers travel as integers


class FhProblem(C.Structure):
    _fields_ = [("op", C.c_int32), ("use_dct", C.c_int32), ("planes", C.c_int32), ("stride", C.c_int32),
                ("ntaps", C.c_int32), ("m", C.c_int32), ("ldm", C.c_int32), ("halo", C.c_int32),
                ("d", C.c_int64), ("sigma_y2", C.c_double),
                ("tap_dy", c_dp), ("tap_dx", c_dp), ("tap_w", c_dp), ("mask", c_dp),
                ("D", c_dp), ("r", c_dp), ("B", c_dp), ("M", c_dp),
                ("ntaps2", C.c_int32), ("halo2", C.c_int32), ("cg_scipy", C.c_int32), ("fold_sym", C.c_int32),
                ("tap2_dy", c_dp), ("tap2_dx", c_dp), ("tap2_w", c_dp),
                ("fold_fwd_w", c_dp), ("fold_fwd_h", c_dp), ("fold_inv_w", c_dp), ("fold_inv_h", c_dp)]


FH_MAX_BATCH = 16
FH_EINVAL, FH_ESIZE, FH_ESYNC = -1, -2, -3  # include/fh_hip.h


class FhBatch(C.Structure):
    _fields_ = [("nimg", C.c_int32), ("pad", C.c_int32), ("D", c_dp * FH_MAX_BATCH), ("r", c_dp * FH_MAX_BATCH),
                ("B", c_dp * FH_MAX_BATCH), ("M", c_dp * FH_MAX_BATCH), ("mask", c_dp * FH_MAX_BATCH)]


class FhCovState(C.Structure):
    _fields_ = [("d", C.c_int64), ("m_c", C.c_int32), ("m_h", C.c_int32), ("ldm", C.c_int32), ("ldg", C.c_int32),
                ("project", C.c_int32), ("use_dct", C.c_int32), ("D", c_dp * 4), ("r", c_dp * 4), ("M", c_dp * 4),
                ("Bc", c_dp), ("Bh", c_dp), ("G", c_dp), ("scal", c_dp), ("t0", c_dp), ("t1", c_dp), ("t2", c_dp)]


class FhGnEpilogue(C.Structure):
    _fields_ = [("partial", c_dp), ("x", c_dp), ("tab", c_dp), ("mode", C.c_int32), ("act", C.c_int32), ("in_amax", c_dp)]


class FhCgInfo(C.Structure):
    _fields_ = [("niter", C.c_int32), ("optimal", C.c_int32), ("residual_norm", C.c_double),
                ("b_norm", C.c_double)]


_SIGS = {
    "fh_version": ([], C.c_int),
    "fh_context_create": ([C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int], C.c_int),
    "fh_context_destroy": ([C.c_void_p], C.c_int),
    "fh_context_set_exclusive": ([C.c_void_p, C.c_int], C.c_int),
    "fh_context_status": ([C.c_void_p, C.c_void_p], C.c_int),
    "fh_debug_read_stamps": ([C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int, C.c_void_p], C.c_int),
    "fh_dct2d": ([C.c_void_p, c_dp, c_dp, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_rep_apply": ([C.c_void_p, c_dp, c_dp, c_dp, c_dp, C.c_int, c_dp, c_dp, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "fh_rep_apply_batched": ([C.c_void_p, C.POINTER(FhBatch), C.c_int, c_dp, c_dp, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "fh_rep_invert": ([C.c_void_p, c_dp, c_dp, c_dp, C.c_double, c_dp, c_dp, c_dp, C.c_int, C.c_int64, C.c_int,
                       C.c_void_p], C.c_int),
    "fh_space_prep": ([C.c_void_p, c_dp, C.c_double, c_dp, c_dp, c_dp, C.c_int64, C.c_void_p], C.c_int),
    "fh_dot": ([C.c_void_p, c_dp, c_dp, c_dp, C.c_int, C.c_int64, C.c_void_p], C.c_int),
    "fh_space_commit": ([C.c_void_p, c_dp, c_dp, C.c_double, C.c_double, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp,
                         c_dp, c_dp, c_dp, C.c_int, C.c_int64, C.c_void_p], C.c_int),
    "fh_space_commit_dev": ([C.c_void_p, c_dp, c_dp, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                             c_dp, C.c_int, C.c_int, c_dp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p], C.c_int),
    "fh_cov_time_update": ([C.c_void_p, C.POINTER(FhCovState), c_dp, c_dp, C.c_double, C.c_double, C.c_double, C.c_int,
                            c_dp, c_dp, c_dp, c_dp, C.c_void_p], C.c_int),
    "fh_cov_space_update": ([C.c_void_p, C.POINTER(FhCovState), c_dp, c_dp, C.c_double, c_dp, c_dp, C.c_void_p], C.c_int),
    "fh_cov_time_update_batched": ([C.c_void_p, C.c_int, C.POINTER(FhCovState), c_dp, c_dp, C.c_double, C.c_double, C.c_double,
                                    c_dp, c_dp, c_dp, C.c_void_p], C.c_int),
    "fh_cov_space_update_batched": ([C.c_void_p, C.c_int, C.POINTER(FhCovState), c_dp, c_dp, C.c_double, c_dp, c_dp, c_dp,
                                     C.c_void_p], C.c_int),
    "fh_woodbury_inner": ([C.c_void_p, c_dp, C.c_int, c_dp, C.c_int, c_dp, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_axpby": ([C.c_double, c_dp, C.c_double, c_dp, c_dp, C.c_int64, C.c_void_p], C.c_int),
    "fh_read_scalars": ([C.c_void_p, c_dp, C.POINTER(C.c_double), C.c_int, C.c_void_p], C.c_int),
    "fh_conv_circ": ([C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_void_p], C.c_int),
    "fh_amm": ([C.c_void_p, C.POINTER(FhProblem), c_dp, c_dp, C.c_void_p], C.c_int),
    "fh_dense_matvec_scratch_doubles": ([C.c_int, C.c_int64], C.c_int64),
    "fh_dense_matvec": ([c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_void_p], C.c_int),
    "fh_dense_rank2": ([c_dp, c_dp, C.c_int, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_double, C.c_double,
                        C.c_void_p], C.c_int),
    "fh_conv2d_nhwc": ([c_dp, c_dp, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 10 + [C.c_void_p], C.c_int),
    "fh_conv2d_x6_nhwc": ([c_dp, c_dp, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 10 + [C.c_void_p], C.c_int),
    "fh_conv2d_x6_nhwc_gn": ([c_dp, c_dp, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 10 + [C.POINTER(FhGnEpilogue), C.c_void_p],
                             C.c_int),
    "fh_conv2d_x6_gn_chunks": ([C.c_int] * 10, C.c_int),
    "fh_groupnorm_finalize": ([c_dp, c_dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p], C.c_int),
    "fh_groupnorm_bwd_table": ([c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int, c_dp, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_groupnorm_bwd_apply": ([c_dp] * 8 + [C.c_int, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_groupnorm_bwd_sums": ([c_dp] * 7 + [C.c_int, c_dp, c_dp] + [C.c_int] * 4 + [C.c_void_p], C.c_int),
    "fh_groupnorm_bwd_apply_ex": ([c_dp] * 8 + [C.c_int, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 5 + [c_dp, C.c_void_p], C.c_int),
    "fh_conv2d_splitk": ([C.c_int] * 7, C.c_int),
    "fh_unet_set_precision": ([C.c_int], C.c_int),
    "fh_absmax_f32": ([c_dp, C.c_int64, c_dp, C.c_void_p], C.c_int),
    "fh_groupnorm_table": ([c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int, c_dp, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_conv2d_x6_norm_supported": ([C.c_int] * 5, C.c_int),
    "fh_conv2d_x6_norm_nhwc": ([c_dp, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_conv2d_x6_norm_nhwc_gn": ([c_dp, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 5
                                  + [C.POINTER(FhGnEpilogue), C.c_void_p], C.c_int),
    "fh_conv3x3_thin_nhwc": ([c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_conv3x3_wino_nhwc": ([c_dp, c_dp, c_dp, c_dp, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_bgemm_f32": ([c_dp, c_dp, c_dp] + [C.c_int] * 10 + [C.c_int64] * 6 + [C.c_float, C.c_void_p], C.c_int),
    "fh_groupnorm_scratch_doubles": ([C.c_int, C.c_int], C.c_int64),
    "fh_groupnorm_stats": ([c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_groupnorm_apply": ([c_dp] * 6 + [C.c_int, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_groupnorm_bwd": ([c_dp] * 7 + [C.c_int, c_dp, c_dp, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_attention_supported": ([C.c_int] * 3, C.c_int),
    "fh_attention_fwd": ([c_dp, c_dp, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_attention_bwd": ([c_dp] * 6 + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_softmax_rows": ([c_dp, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "fh_softmax_bwd_rows": ([c_dp, c_dp, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "fh_resample2x": ([c_dp, c_dp] + [C.c_int] * 5 + [C.c_void_p], C.c_int),
    "fh_concat_channels": ([c_dp, c_dp, c_dp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_layout_nchw_nhwc": ([c_dp, c_dp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "fh_add_f32": ([c_dp, c_dp, c_dp, C.c_int64, C.c_void_p], C.c_int),
    "fh_metrics_scratch_doubles": ([C.c_int] * 4, C.c_int64),
    "fh_metrics_u8": ([c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_void_p], C.c_int),
    "fh_cg_solve_batched": ([C.c_void_p, C.POINTER(FhProblem), C.POINTER(FhBatch), c_dp, c_dp, C.POINTER(C.c_double),
                             C.c_double, C.c_int, C.POINTER(FhCgInfo), C.c_void_p], C.c_int),
    "fh_cg_solve": ([C.c_void_p, C.POINTER(FhProblem), c_dp, c_dp, C.c_double, C.c_double, C.c_int,
                     C.POINTER(FhCgInfo), C.c_void_p], C.c_int),
}

_lib = None


class FhError(RuntimeError):
    pass


def load():
    """Load libfh_hip.so and declare every prototype.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FhError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what):
    if rc != 0:
        raise FhError(f"{what} failed with code {rc}" + (" (hipError)" if rc > 0 else " (bad argument / size)"))


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


class Context:
    """fh_context for one (image side, planes, column capacity) geometry on the current device."""

    _cache = {}

    def __init__(self, S, planes, m_cap):
        self.lib = load()
        self.S, self.planes, self.m_cap = S, planes, m_cap
        h = C.c_void_p()
        check(self.lib.fh_context_create(C.byref(h), S, planes, m_cap), "fh_context_create")
        self.h = h

    @classmethod
    def get(cls, S, planes, m_cap, slot=0):
        """One context per (device, geometry, slot).  A context owns scratch buffers, so concurrent users (one image
        per HIP stream / host thread) take different slots."""
        key = (torch.cuda.current_device(), S, planes, m_cap, slot)
        if key not in cls._cache:
            cls._cache[key] = cls(S, planes, m_cap)
        return cls._cache[key]

    def set_exclusive(self, flag):
        """Declare that no other grid-synchronising kernel shares the GPU with this context's stream (include/fh_hip.h):
        the covariance apply then reads the factor base once instead of twice."""
        check(self.lib.fh_context_set_exclusive(self.h, int(flag)), "fh_context_set_exclusive")

    def status(self):
        """Raises if a single-sweep covariance apply of this context timed out since the last call."""
        check(self.lib.fh_context_status(self.h, stream()), "fh_context_status")

    # thin typed wrappers ------------------------------------------------------------------
    def dct2d(self, x, out=None, inverse=False):
        assert x.dtype == torch.float64 and x.shape[-1] == self.S and x.shape[-2] == self.S
        out = torch.empty_like(x) if out is None else out
        planes = x.numel() // (self.S * self.S)
        check(self.lib.fh_dct2d(self.h, ptr(x), ptr(out), planes, int(inverse), stream()), "fh_dct2d")
        return out

    def rep_apply(self, D, r, B, M, z, out, m):
        check(self.lib.fh_rep_apply(self.h, ptr(D), ptr(r) if m else None, ptr(B) if m else None,
                                    ptr(M) if m else None, M.shape[1] if m else 0, ptr(z), ptr(out), D.numel(), m,
                                    stream()), "fh_rep_apply")
        return out

    def rep_invert(self, Dx, rx, B, shift, Dy, ry, G, m):
        # the row scales travel also for an empty factor: the closed-form C^-1 update of fh_cov_space_update relies on
        # r_inverse = r / D from the first pair on
        check(self.lib.fh_rep_invert(self.h, ptr(Dx), ptr(rx), ptr(B) if m else None, float(shift),
                                     ptr(Dy), ptr(ry), ptr(G) if m else None,
                                     G.shape[1] if m else 0, Dx.numel(), m, stream()), "fh_rep_invert")

    def read_scalars(self, scal, k):
        buf = (C.c_double * k)()
        check(self.lib.fh_read_scalars(self.h, ptr(scal), buf, k, stream()), "fh_read_scalars")
        return list(buf)

    def axpby(self, alpha, a, beta, b, out):
        check(self.lib.fh_axpby(float(alpha), ptr(a), float(beta), ptr(b), ptr(out), a.numel(), stream()), "fh_axpby")
        return out

    def conv(self, x, out, taps, planes, stride=1, adjoint=False):
        check(self.lib.fh_conv_circ(self.h, ptr(x), ptr(out), ptr(taps.dy), ptr(taps.dx), ptr(taps.w), taps.n,
                                    taps.halo, planes, stride, int(adjoint), stream()), "fh_conv_circ")
        return out
