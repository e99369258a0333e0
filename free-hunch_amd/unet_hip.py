"""Execution of the UNet step list on the gfx950 kernels of libfh_hip.so (forward and input-VJP).

Activations are NHWC float32 tensors in HBM (torch only owns the memory).  The whole network is ONE
`torch.autograd.Function`: the forward pass records what the input-gradient needs (GroupNorm inputs and
statistics, attention probabilities, qkv) on a tape, the backward pass walks the tape in reverse with the
dgrad / GroupNorm-backward / attention-backward kernels.  Weight gradients are never formed - the Free Hunch
path only needs J^T v with respect to the image (conditioning_mechanisms.py:280).

Weights are repacked once per device: conv weights [Cout][Cin][kh][kw] -> [Cout][taps][Cin] for the forward
implicit GEMM and -> [Cin][taps flipped][Cout] for the input gradient (Cin / Cout zero-padded to a multiple
of 32 where they are the GEMM K dimension).
"""
from __future__ import annotations

import math
import os
import threading

import torch
import torch.nn.functional as F

import ctypes as C

from . import _lib

_K = 32  # K granularity of fh_conv2d_nhwc


def _pad_to(n, m):
    return (n + m - 1) // m * m


def _wino(w):
    """[rows][ky][kx][K] -> [4][rows][ky][K]: u = (w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2) over kx"""
    w0, w1, w2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    return torch.stack([w0, (w0 + w1 + w2) / 2, (w0 - w1 + w2) / 2, w2], 0).contiguous()


def _use_wino(N, H, W, K, rows_out):
    """Measured on MI355X (profiles/tools/bench_wino.py): the fused F(2,3) kernel is 1.23-1.25x the direct kernel when a full
    wave of 256-tile workgroups covers the chip (1.11x at K = Cin = 128), ~1.14x with the 128-tile variant at K >= 256;
    smaller grids stay on the direct kernel (or split-K).  Mirrors the variant choice of fh_conv3x3_wino_nhwc."""
    if W % 2 or K % 16 or K < 128:
        return False
    mt, nb = N * H * (W // 2), -(-rows_out // 64)
    if -(-mt // 256) * nb >= 224:
        return True
    return K >= 256 and -(-mt // 128) * nb >= 256


def _split3(w):
    """fp32 [rows][taps][K] -> bf16 [3 planes][taps][K/32][rows][32] with w == p0 + p1 + p2 exactly (fh_conv2d_x6_nhwc)."""
    h = w.to(torch.bfloat16)
    r = w - h.float()
    m = r.to(torch.bfloat16)
    lo = (r - m.float()).to(torch.bfloat16)
    rows, taps, k = w.shape
    return torch.stack([h, m, lo]).reshape(3, rows, taps, k // _K, _K).permute(0, 2, 3, 1, 4).contiguous()


def _half_plane(w):
    """fp32 [rows][taps][K] -> the half-precision bit patterns in the split-bf16 kernels' weight layout, ONE plane
    [1][taps][K/32][rows][32] (int16 storage; the fp16 mode of fh_unet_set_precision reads it through the f16 MFMA)."""
    rows, taps, k = w.shape
    return w.to(torch.float16).view(torch.int16).reshape(1, rows, taps, k // _K, _K).permute(0, 2, 3, 1, 4).contiguous()


_AMAX_SLOTS = 16  # FH_AMAX_SLOTS of include/fh_hip.h


def _half_split_planes(w):
    """fp32 [rows][taps][K] -> the weight operand of the half-split mode (fh_unet_set_precision(4)): int16 bit patterns of
    h = rn_half(w 2^k), m = rn_half(w 2^k - h) as [2 planes][taps][K/32][rows][32], followed by ONE float32 = 2^-k (two int16
    slots).  k puts the largest |w| into [2^12, 2^13): every weight above 2^-16 of the largest keeps h + m == w 2^k to one
    fp32 ulp (m stays a normal half)."""
    rows, taps, k = w.shape
    amax = float(w.abs().max())
    kk = 12 - math.floor(math.log2(amax)) if amax > 0 else 0
    ws = w * (2.0 ** kk)
    h = ws.to(torch.float16)
    m = (ws - h.float()).to(torch.float16)
    planes = torch.stack([h, m]).view(torch.int16).reshape(2, rows, taps, k // _K, _K).permute(0, 2, 3, 1, 4).contiguous()
    tail = torch.tensor([2.0 ** -kk], dtype=torch.float32, device=w.device).view(torch.int16)
    return torch.cat([planes.flatten(), tail])


def _use_x6(N, H, W, rows_out):
    """Measured on MI355X (profiles/tools/bench_x6.py): the split-bf16 kernel is 1.2-1.4x the fp32-MFMA kernel (1.05-1.1x the
    Winograd one) on the large layers and, with the same deterministic split-K, 1.0-1.2x on the 8 x 8 ... 32 x 32 grids;
    only the thin outputs (<= 64 rows: the 3 / 6 channel image convolutions) stay on the fp32 kernel."""
    return rows_out > 64


# conv arithmetic: "x6" = exact-split bf16 MFMA (fp32 accuracy, default), "wino" = fp32 MFMA + F(2,3), "f32" = fp32 MFMA only
CONV_MODE = os.environ.get("FH_CONV_MODE", "x6")
assert CONV_MODE in ("x6", "wino", "f32"), CONV_MODE


class _Conv:
    def __init__(self, w, b):
        co, ci = w.shape[0], w.shape[1]
        kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)  # conv1d qkv / proj_out are 1x1
        w4 = w.reshape(co, ci, kh, kw).float()
        self.co, self.ci, self.kh, self.kw = co, ci, kh, kw
        self.ci_p, self.co_p = _pad_to(ci, _K), _pad_to(co, _K)
        wf = torch.zeros(co, kh * kw, self.ci_p, dtype=torch.float32, device=w.device)
        wf[:, :, :ci] = w4.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
        wd = torch.zeros(ci, kh * kw, self.co_p, dtype=torch.float32, device=w.device)
        wd[:, :, :co] = w4.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, kh * kw, co)
        self.wf, self.wd, self.b = wf.contiguous(), wd.contiguous(), b.float().contiguous()
        # F(2,3) Winograd copies along kx, [4][rows][3 ky][K]: used for the large 3x3 layers (fh_conv3x3_wino_nhwc)
        self.wu_f = self.wu_d = self.wx_f = self.wx_d = None
        if kh == 3 and kw == 3 and CONV_MODE == "wino":
            self.wu_f = _wino(wf.reshape(co, 3, 3, self.ci_p))
            self.wu_d = _wino(wd.reshape(ci, 3, 3, self.co_p))
        self.wh_f = self.wh_d = None  # half-precision planes, built when the fp16 mode is first used
        self.ws_f = self.ws_d = None  # half-split planes (precision mode 4), built on first use
        if CONV_MODE == "x6":  # exact bf16 split of both copies (6 bytes per weight)
            self.wx_f, self.wx_d = _split3(self.wf), _split3(self.wd)

    def planes(self, fwd, mode):
        """the weight operand of the split-bf16 kernels for fh_unet_set_precision(mode): the three bf16 planes (modes 0-2),
        the half-precision plane (3) or the two half-split planes + scale (4)"""
        if mode == 4:
            if self.ws_f is None:
                self.ws_f, self.ws_d = _half_split_planes(self.wf), _half_split_planes(self.wd)
            return self.ws_f if fwd else self.ws_d
        if mode != 3:
            return self.wx_f if fwd else self.wx_d
        if self.wh_f is None:
            self.wh_f, self.wh_d = _half_plane(self.wf), _half_plane(self.wd)
        return self.wh_f if fwd else self.wh_d


class HipOps:
    def __init__(self, cfg, P):
        self.cfg = cfg
        self.lib = _lib.load()
        self.P = P
        self.bf16 = 0  # fh_unet_set_precision code of the torso (UNetModel.set_dtype): 0 fp32-exact, 1 bf16, 2 bf16x3, 3 fp16
        self.conv = {}
        for k, v in P.items():
            if k.endswith(".weight") and v.dim() >= 3:
                name = k[: -len(".weight")]
                self.conv[name] = _Conv(v.detach(), P[name + ".bias"].detach())
        self._emb_cache = {}
        self._tls = threading.local()  # emb_key: the timestep tuple of the call running on THIS host thread

    # ---------------------------------------------------------------- kernel wrappers
    def _launch_mode(self, c, stride=1, x=None):
        """fh_unet_set_precision code of ONE split-kernel launch.  The half-split mode (4) covers the 3 x 3 / stride 1
        layers (97 % of the matrix work) and the 1 x 1 ones whose input `x` arrives with its magnitude (`_fh_amax`, left by the
        GroupNorm-backward pass that wrote it); other 1 x 1 and strided convolutions keep the exact bf16 split, where the
        extra pass for the input's magnitude would cost what the cheaper products save."""
        if self.bf16 != 4:
            return self.bf16
        m = 4 if (stride == 1 and ((c.kh == 3 and c.kw == 3) or getattr(x, "_fh_amax", None) is not None)) else 0
        self.lib.fh_unet_set_precision(m)  # (per host thread)
        return m

    def _amax_slot(self):
        """[2][FH_AMAX_SLOTS] zeroed device floats out of a pool that one fill serves for a whole backward pass"""
        pool = getattr(self._tls, "amax_pool", None)
        if pool is None or pool[1] >= pool[0].shape[0]:
            dev = self.P[next(iter(self.P))].device
            pool = self._tls.amax_pool = [torch.zeros(256, 2, _AMAX_SLOTS, dtype=torch.float32, device=dev), 0]
        pool[1] += 1
        return pool[0][pool[1] - 1]

    def _amax(self, x):
        pre = getattr(x, "_fh_amax", None)  # left by the GroupNorm-backward pass that wrote x
        if pre is not None:
            return pre
        out = torch.empty(_AMAX_SLOTS, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.fh_absmax_f32(x.data_ptr(), x.numel(), out.data_ptr(), _lib.stream()), "fh_absmax_f32")
        return out

    def _conv(self, name, x, res=None, bias_override=None, stride=1):
        c = self.conv[name]
        N, H, W, Ci = x.shape
        assert Ci == c.ci_p, (name, x.shape, c.ci_p)
        pad = c.kh // 2
        b = c.b if bias_override is None else bias_override
        if stride != 1:  # Downsample(use_conv=True), openai_unet.py:131: 3x3, stride 2, padding 1
            Ho, Wo = (H + 2 * pad - c.kh) // stride + 1, (W + 2 * pad - c.kw) // stride + 1
            out = torch.empty(N, Ho, Wo, c.co, dtype=torch.float32, device=x.device)
            ks = self.lib.fh_conv2d_splitk(N, Ho, Wo, Ci, c.co, c.kh, c.kw)
            ws = torch.empty(ks, N * Ho * Wo, c.co, dtype=torch.float32, device=x.device) if ks > 1 else None
            fn, wgt = ((self.lib.fh_conv2d_x6_nhwc, c.planes(True, self._launch_mode(c, stride)))
                       if c.wx_f is not None and _use_x6(N, Ho, Wo, c.co) else (self.lib.fh_conv2d_nhwc, c.wf))
            _lib.check(fn(x.data_ptr(), wgt.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(),
                          out.data_ptr(), None if ws is None else ws.data_ptr(), ks, N, H, W, Ci, c.co, c.kh, c.kw, pad,
                          stride, _lib.stream()), "fh_conv2d(stride)")
            return out
        out = torch.empty(N, H, W, c.co, dtype=torch.float32, device=x.device)
        if c.wu_f is not None and _use_wino(N, H, W, Ci, c.co):
            _lib.check(self.lib.fh_conv3x3_wino_nhwc(x.data_ptr(), c.wu_f.data_ptr(), b.data_ptr(),
                                                     None if res is None else res.data_ptr(), out.data_ptr(), N, H, W,
                                                     Ci, c.co, _lib.stream()), "fh_conv3x3_wino_nhwc")
            return out
        if c.co <= 8 and c.kh == 3 and c.kw == 3 and res is None and H * W >= 4096:
            _lib.check(self.lib.fh_conv3x3_thin_nhwc(x.data_ptr(), c.wf.data_ptr(), b.data_ptr(), out.data_ptr(), N, H, W,
                                                     Ci, c.co, _lib.stream()), "fh_conv3x3_thin_nhwc")
            return out
        ks = self.lib.fh_conv2d_splitk(N, H, W, Ci, c.co, c.kh, c.kw)
        ws = torch.empty(ks, N * H * W, c.co, dtype=torch.float32, device=x.device) if ks > 1 else None
        if c.wx_f is not None and _use_x6(N, H, W, c.co):
            # group-sum epilogue: the statistics of a GroupNorm applied to this output come out of the convolution itself
            m = self._launch_mode(c)
            amax = self._amax(x) if m == 4 else None
            epi, keep = self._epilogue(ks, N, H, W, Ci, c.co, c.kh, c.kw, pad, 0, amax=amax)
            _lib.check(self.lib.fh_conv2d_x6_nhwc_gn(x.data_ptr(), c.planes(True, m).data_ptr(), b.data_ptr(),
                                                     None if res is None else res.data_ptr(), out.data_ptr(),
                                                     None if ws is None else ws.data_ptr(), ks, N, H, W, Ci, c.co, c.kh,
                                                     c.kw, pad, 1, epi, _lib.stream()), "fh_conv2d_x6_nhwc")
            if keep is not None:
                out._fh_gn = keep
            return out
        _lib.check(self.lib.fh_conv2d_nhwc(x.data_ptr(), c.wf.data_ptr(), b.data_ptr(),
                                           None if res is None else res.data_ptr(), out.data_ptr(),
                                           None if ws is None else ws.data_ptr(), ks, N, H, W, Ci, c.co,
                                           c.kh, c.kw, pad, 1, _lib.stream()), "fh_conv2d_nhwc")
        return out

    def _dgrad(self, name, g, res=None, gn=None):
        """Input gradient of convolution `name`.  `gn` = (norm name, x, stats, act, scale, shift) when the result is the
        dL/dy of that GroupNorm: where the launch supports it the two sums of its backward come out of the convolution's
        epilogue (returned as `out._fh_gn_sums` for `_gn_bwd`)."""
        c = self.conv[name]
        N, H, W, Co = g.shape
        if Co != c.co_p:  # only the 6-channel output conv: pad the cotangent to the K granularity
            gp = torch.zeros(N, H, W, c.co_p, dtype=torch.float32, device=g.device)
            gp[..., :Co] = g
            g = gp
        out = torch.empty(N, H, W, c.ci, dtype=torch.float32, device=g.device)
        if c.wu_d is not None and _use_wino(N, H, W, c.co_p, c.ci):
            _lib.check(self.lib.fh_conv3x3_wino_nhwc(g.data_ptr(), c.wu_d.data_ptr(), None,
                                                     None if res is None else res.data_ptr(), out.data_ptr(), N, H, W,
                                                     c.co_p, c.ci, _lib.stream()), "fh_conv3x3_wino_nhwc(dgrad)")
            return out
        if c.ci <= 8 and c.kh == 3 and c.kw == 3 and res is None and H * W >= 4096:
            _lib.check(self.lib.fh_conv3x3_thin_nhwc(g.data_ptr(), c.wd.data_ptr(), None, out.data_ptr(), N, H, W, c.co_p,
                                                     c.ci, _lib.stream()), "fh_conv3x3_thin_nhwc(dgrad)")
            return out
        ks = self.lib.fh_conv2d_splitk(N, H, W, c.co_p, c.ci, c.kh, c.kw)
        ws = torch.empty(ks, N * H * W, c.ci, dtype=torch.float32, device=g.device) if ks > 1 else None
        if c.wx_d is not None and _use_x6(N, H, W, c.ci):
            m = self._launch_mode(c, 1, g)
            amax = self._amax(g) if m == 4 else None
            epi, keep = self._epilogue(ks, N, H, W, c.co_p, c.ci, c.kh, c.kw, c.kh // 2, 1,
                                       gn if gn is not None and gn[1].shape == out.shape else None, amax=amax)
            _lib.check(self.lib.fh_conv2d_x6_nhwc_gn(g.data_ptr(), c.planes(False, m).data_ptr(), None,
                                                     None if res is None else res.data_ptr(), out.data_ptr(),
                                                     None if ws is None else ws.data_ptr(), ks, N, H, W, c.co_p, c.ci, c.kh,
                                                     c.kw, c.kh // 2, 1, epi, _lib.stream()), "fh_conv2d_x6_nhwc(dgrad)")
            if keep is not None:
                partial, chunks = keep[0], keep[1]
                sums = torch.empty(N, 32, 2, dtype=torch.float32, device=g.device)
                _lib.check(self.lib.fh_groupnorm_finalize(partial.data_ptr(), sums.data_ptr(), N, chunks,
                                                          float(H * W * (c.ci // 32)), 1, _lib.stream()), "gn_finalize(1)")
                out._fh_gn_sums = sums
            return out
        _lib.check(self.lib.fh_conv2d_nhwc(g.data_ptr(), c.wd.data_ptr(), None,
                                           None if res is None else res.data_ptr(), out.data_ptr(),
                                           None if ws is None else ws.data_ptr(), ks, N, H, W, c.co_p,
                                           c.ci, c.kh, c.kw, c.kh // 2, 1, _lib.stream()), "fh_conv2d_nhwc(dgrad)")
        return out

    def _epilogue(self, ks, N, H, W, Ci, Co, kh, kw, pad, mode, gn=None, amax=None):
        """fh_gn_epilogue for the split-bf16 convolution launch of this layer, or (None, None) where the launch has none
        (split-K, thin output, ...).  mode 0: statistics of the output; mode 1: backward sums of the GroupNorm `gn` (None =
        no group sums wanted).  `amax` (half-split mode): the input's magnitude rides in the same struct; the second
        return value is None when the launch leaves no group partials."""
        chunks = 0
        if os.environ.get("FH_GN_EPILOGUE", "1") != "0" and Co % 32 == 0 and not (mode == 1 and gn is None):
            chunks = self.lib.fh_conv2d_x6_gn_chunks(ks, N, H, W, Ci, Co, kh, kw, pad, 1)
        if chunks <= 0:
            if amax is None:
                return None, None
            e = _lib.FhGnEpilogue()
            e.partial, e.in_amax = None, amax.data_ptr()
            return C.byref(e), None
        dev = self.P[next(iter(self.P))].device
        partial = torch.empty(N * chunks * 64, dtype=torch.float64, device=dev)
        e = _lib.FhGnEpilogue()
        e.partial, e.mode, e.act = partial.data_ptr(), mode, 0
        e.in_amax = None if amax is None else amax.data_ptr()
        keep = [partial, chunks]
        if mode == 1:
            gn_name, x, stats, act, scale, shift = gn
            table = torch.empty(N, 5, Co, dtype=torch.float32, device=dev)
            ss = 0 if scale is None else scale.stride(0)
            _lib.check(self.lib.fh_groupnorm_bwd_table(
                stats.data_ptr(), self.P[gn_name + ".weight"].data_ptr(), self.P[gn_name + ".bias"].data_ptr(),
                None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), ss,
                table.data_ptr(), N, Co, _lib.stream()), "gn_bwd_table")
            e.x, e.tab, e.act = x.data_ptr(), table.data_ptr(), int(act)
            keep.append(table)
        return C.byref(e), keep

    def _gn_stats(self, x):
        N, H, W, C_ = x.shape
        pre = getattr(x, "_fh_gn", None)
        if pre is not None:  # block partials left by the producing convolution's epilogue
            stats = torch.empty(N, 32, 2, dtype=torch.float32, device=x.device)
            _lib.check(self.lib.fh_groupnorm_finalize(pre[0].data_ptr(), stats.data_ptr(), N, pre[1],
                                                      float(H * W * (C_ // 32)), 0, _lib.stream()), "gn_finalize(0)")
            return stats
        return self._gn_stats_pass(x)

    def _gn_stats_pass(self, x):
        N, H, W, C = x.shape
        stats = torch.empty(N, 32, 2, dtype=torch.float32, device=x.device)
        scratch = torch.empty(self.lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=x.device)
        _lib.check(self.lib.fh_groupnorm_stats(x.data_ptr(), stats.data_ptr(), scratch.data_ptr(), N, H * W, C,
                                               _lib.stream()), "gn_stats")
        return stats

    def _gn_apply(self, name, x, stats, act, scale=None, shift=None):
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        ss = 0 if scale is None else scale.stride(0)
        _lib.check(self.lib.fh_groupnorm_apply(
            x.data_ptr(), stats.data_ptr(), self.P[name + ".weight"].data_ptr(), self.P[name + ".bias"].data_ptr(),
            None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), ss,
            y.data_ptr(), N, H * W, C, int(act), _lib.stream()), "gn_apply")
        return y

    def _gn(self, name, x, act, scale=None, shift=None):
        stats = self._gn_stats(x)
        return self._gn_apply(name, x, stats, act, scale, shift), stats

    def _gn_conv(self, gn_name, conv_name, x, act, scale=None, shift=None, res=None):
        """conv3x3(act(GroupNorm(x))) - the in_layers / out_layers pattern of a ResBlock.  Where the row-reuse split-bf16
        kernel applies, the normalisation is folded into its tile staging (fh_conv2d_x6_norm_nhwc): the normalised tensor
        is never written.  Returns (conv output, GroupNorm statistics)."""
        c = self.conv[conv_name]
        N, H, W, Ci = x.shape
        stats = self._gn_stats(x)
        if (c.wx_f is not None and c.kh == 3 and c.kw == 3 and Ci == c.ci_p and os.environ.get("FH_GN_FUSE", "1") != "0"
                and self.lib.fh_conv2d_x6_norm_supported(N, H, W, Ci, c.co)):
            table = torch.empty(N, 2, Ci, dtype=torch.float32, device=x.device)
            ss = 0 if scale is None else scale.stride(0)
            _lib.check(self.lib.fh_groupnorm_table(
                stats.data_ptr(), self.P[gn_name + ".weight"].data_ptr(), self.P[gn_name + ".bias"].data_ptr(),
                None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), ss,
                table.data_ptr(), N, Ci, _lib.stream()), "gn_table")
            out = torch.empty(N, H, W, c.co, dtype=torch.float32, device=x.device)
            epi, keep = self._epilogue(1, N, H, W, Ci, c.co, 3, 3, 1, 0)
            _lib.check(self.lib.fh_conv2d_x6_norm_nhwc_gn(x.data_ptr(), table.data_ptr(), int(act),
                                                          c.planes(True, self._launch_mode(c)).data_ptr(),
                                                          c.b.data_ptr(), None if res is None else res.data_ptr(),
                                                          out.data_ptr(), N, H, W, Ci, c.co, epi, _lib.stream()),
                       "fh_conv2d_x6_norm_nhwc")
            if keep is not None:
                out._fh_gn = keep
            return out, stats
        return self._conv(conv_name, self._gn_apply(gn_name, x, stats, act, scale, shift), res=res), stats

    def _gn_bwd(self, name, x, stats, dy, act, scale=None, shift=None, accumulate_into=None, add2=None, split=None):
        """dL/dx of GroupNorm(+scale-shift)(+SiLU) given dL/dy.  `accumulate_into`: the gradient arriving over the block's
        skip path is added (in place unless `split`); `add2`: one more addend (the gradient of a U-Net skip tensor forked at
        this point); `split` = (Ca, Cb): the result is returned as two tensors [.., Ca], [.., Cb] (x was a channel concat)."""
        N, H, W, C_ = x.shape
        ss = 0 if scale is None else scale.stride(0)
        gp = (self.P[name + ".weight"].data_ptr(), self.P[name + ".bias"].data_ptr(),
              None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), ss)
        sums = getattr(dy, "_fh_gn_sums", None)  # left by the producing input-gradient convolution's epilogue
        if sums is None:
            sums = torch.empty(N, 32, 2, dtype=torch.float32, device=x.device)
            scratch = torch.empty(self.lib.fh_groupnorm_scratch_doubles(N, H * W), dtype=torch.float64, device=x.device)
            _lib.check(self.lib.fh_groupnorm_bwd_sums(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), *gp, sums.data_ptr(),
                                                      scratch.data_ptr(), N, H * W, C_, int(act), _lib.stream()), "gn_bwd_sums")
        acc = accumulate_into
        if split is not None:
            dx = torch.empty(N, H, W, split[0], dtype=torch.float32, device=x.device)
            dx2 = torch.empty(N, H, W, split[1], dtype=torch.float32, device=x.device)
        else:
            dx, dx2 = (acc if acc is not None else torch.empty_like(x)), None
        amax = self._amax_slot() if self.bf16 == 4 else None  # the magnitudes of dx, dx2 for the half-split dgrad that follows
        _lib.check(self.lib.fh_groupnorm_bwd_apply_ex(
            x.data_ptr(), dy.data_ptr(), stats.data_ptr(), sums.data_ptr(), *gp, None if acc is None else acc.data_ptr(),
            None if add2 is None else add2.data_ptr(), dx.data_ptr(), None if dx2 is None else dx2.data_ptr(),
            0 if split is None else split[0], N, H * W, C_, int(act), None if amax is None else amax.data_ptr(),
            _lib.stream()), "gn_bwd_apply")
        if amax is not None:
            dx._fh_amax = amax[0]
            if dx2 is not None:
                dx2._fh_amax = amax[1]
        return dx if split is None else (dx, dx2)

    def _resample(self, x, mode):
        N, H, W, C = x.shape
        hs, ws = (H // 2, W // 2) if mode in (0, 3) else (H, W)  # modes 1, 2, 4 go small -> big
        oh, ow = (hs, ws) if mode in (0, 3) else (2 * H, 2 * W)
        out = torch.empty(N, oh, ow, C, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.fh_resample2x(x.data_ptr(), out.data_ptr(), N, hs, ws, C, mode, _lib.stream()), "resample")
        return out

    def _add(self, a, b):
        out = torch.empty_like(a)
        _lib.check(self.lib.fh_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _lib.stream()), "add")
        return out

    def _bgemm(self, A, B, Cm, M, N, K, lda, ldb, ldc, ta, tb, batch, inner, sA, sB, sC, alpha=1.0):
        _lib.check(self.lib.fh_bgemm_f32(A, B, Cm, M, N, K, lda, ldb, ldc, ta, tb, batch, inner, sA[0], sA[1], sB[0],
                                         sB[1], sC[0], sC[1], float(alpha), _lib.stream()), "fh_bgemm_f32")

    # ---------------------------------------------------------------- timestep-embedding MLPs (depend on sigma only)
    def emb_out(self, prefix, emb):
        ek = getattr(self._tls, "emb_key", None)
        key = (ek, prefix)
        e = self._emb_cache.get(key) if ek is not None else None
        if e is None:
            e = F.linear(F.silu(emb), self.P[prefix + ".emb_layers.1.weight"], self.P[prefix + ".emb_layers.1.bias"])
            e = e.float().contiguous()
            if ek is not None:
                if len(self._emb_cache) > 4096:
                    self._emb_cache.clear()
                self._emb_cache[key] = e
        return e

    # ---------------------------------------------------------------- blocks
    def _res_fwd(self, op, p, x, emb, tape):
        cfg = self.cfg
        e = self.emb_out(p, emb)  # [N, Co] or [N, 2 Co]
        co = self.conv[p + ".in_layers.2"].co
        xs = x
        h0 = None
        if op != "res":  # the resampling sits between the normalisation and the convolution (resblock_updown)
            h0, st0 = self._gn(p + ".in_layers.0", x, act=1)
            mode = 0 if op == "res_down" else 2
            h0, xs = self._resample(h0, mode), self._resample(x, mode)
        if cfg.use_scale_shift_norm:
            if h0 is None:
                h1, st0 = self._gn_conv(p + ".in_layers.0", p + ".in_layers.2", x, act=1)
            else:
                h1 = self._conv(p + ".in_layers.2", h0)
            scale, shift = e[:, :co], e[:, co:]
        else:
            if e.shape[0] != 1 and not bool((e == e[:1]).all()):
                raise NotImplementedError("per-sample timestep embeddings without scale-shift norm")
            if h0 is None:
                h0, st0 = self._gn(p + ".in_layers.0", x, act=1)
            h1 = self._conv(p + ".in_layers.2", h0, bias_override=(self.conv[p + ".in_layers.2"].b + e[0]).contiguous())
            scale = shift = None
        skip = self._conv(p + ".skip_connection", xs) if (p + ".skip_connection") in self.conv else xs
        out, st1 = self._gn_conv(p + ".out_layers.0", p + ".out_layers.3", h1, act=1, scale=scale, shift=shift, res=skip)
        tape.append(("res", op, p, x, st0, h1, st1, scale, shift))
        return out

    def _res_bwd(self, rec, g, add2=None, split=None):
        _, op, p, x, st0, h1, st1, scale, shift = rec
        g_h2 = self._dgrad(p + ".out_layers.3", g, gn=(p + ".out_layers.0", h1, st1, 1, scale, shift))
        g_h1 = self._gn_bwd(p + ".out_layers.0", h1, st1, g_h2, 1, scale, shift)
        # (with a resampling between the norm and the convolution the gradient is not yet the norm's dL/dy)
        g_h0 = self._dgrad(p + ".in_layers.2", g_h1, gn=(p + ".in_layers.0", x, st0, 1, None, None) if op == "res" else None)
        g_xs = self._dgrad(p + ".skip_connection", g) if (p + ".skip_connection") in self.conv else g
        if op == "res_down":
            g_h0, g_xs = self._resample(g_h0, 1), self._resample(g_xs, 1)
        elif op == "res_up":
            g_h0, g_xs = self._resample(g_h0, 3), self._resample(g_xs, 3)
        # (identity skip, no resampling: g_xs aliases g; every read of g is already enqueued on this stream)
        return self._gn_bwd(p + ".in_layers.0", x, st0, g_h0, 1, accumulate_into=g_xs, add2=add2, split=split)

    def _attn_geometry(self, N, T, C, heads):
        ch = C // heads
        if self.cfg.use_new_attention_order:
            hs, qo, ko, vo = ch, 0, C, 2 * C
        else:
            hs, qo, ko, vo = 3 * ch, 0, ch, 2 * ch
        return ch, hs, qo, ko, vo

    def _attn_fwd(self, p, x, heads, tape):
        N, H, W, C = x.shape
        T = H * W
        hn, st = self._gn(p + ".norm", x, act=0)
        qkv = self._conv(p + ".qkv", hn)  # [N,H,W,3C]
        if (os.environ.get("FH_ATTN_FUSED", "1") != "0" and self.lib.fh_attention_supported(T, C, heads)
                and N * heads <= 65535):  # (image, head) pairs ride in gridDim.y
            # one kernel: q k^T -> online softmax -> . v; the tape keeps the output and the log-sum-exp, not the weights
            A = torch.empty(N, H, W, C, dtype=torch.float32, device=x.device)
            lse = torch.empty(N * heads, T, dtype=torch.float32, device=x.device)
            _lib.check(self.lib.fh_attention_fwd(qkv.data_ptr(), A.data_ptr(), lse.data_ptr(), N, T, C, heads,
                                                 int(self.cfg.use_new_attention_order), _lib.stream()), "attention_fwd")
            out = self._conv(p + ".proj_out", A, res=x)
            tape.append(("attn", p, x, st, qkv, (A, lse), heads))
            return out
        ch, hs, qo, ko, vo = self._attn_geometry(N, T, C, heads)
        S = torch.empty(N * heads, T, T, dtype=torch.float32, device=x.device)
        base, fs = qkv.data_ptr(), 4
        s3 = (T * 3 * C, hs)
        self._bgemm(base + qo * fs, base + ko * fs, S.data_ptr(), T, T, ch, 3 * C, 3 * C, T, 0, 0, N * heads, heads,
                    s3, s3, (heads * T * T, T * T), alpha=1.0 / math.sqrt(ch))
        _lib.check(self.lib.fh_softmax_rows(S.data_ptr(), N * heads * T, T, _lib.stream()), "softmax")
        A = torch.empty(N, H, W, C, dtype=torch.float32, device=x.device)
        self._bgemm(S.data_ptr(), base + vo * fs, A.data_ptr(), T, ch, T, T, 3 * C, C, 0, 1, N * heads, heads,
                    (heads * T * T, T * T), s3, (T * C, ch))
        out = self._conv(p + ".proj_out", A, res=x)
        tape.append(("attn", p, x, st, qkv, S, heads))
        return out

    def _attn_bwd(self, rec, g, add2=None):
        _, p, x, st, qkv, S, heads = rec
        N, H, W, C = x.shape
        T = H * W
        ch, hs, qo, ko, vo = self._attn_geometry(N, T, C, heads)
        gA = self._dgrad(p + ".proj_out", g)  # [N,H,W,C]
        dqkv = torch.empty_like(qkv)
        if isinstance(S, tuple):  # fused form: the weights are recomputed from q, k and the log-sum-exp
            A, lse = S
            dsum = torch.empty_like(lse)
            _lib.check(self.lib.fh_attention_bwd(qkv.data_ptr(), A.data_ptr(), gA.data_ptr(), lse.data_ptr(), dsum.data_ptr(),
                                                 dqkv.data_ptr(), N, T, C, heads, int(self.cfg.use_new_attention_order),
                                                 _lib.stream()), "attention_bwd")
            g_hn = self._dgrad(p + ".qkv", dqkv, gn=(p + ".norm", x, st, 0, None, None))
            return self._gn_bwd(p + ".norm", x, st, g_hn, 0, accumulate_into=g, add2=add2)
        base, dbase, fs = qkv.data_ptr(), dqkv.data_ptr(), 4
        s3, sS, sA = (T * 3 * C, hs), (heads * T * T, T * T), (T * C, ch)
        B_ = N * heads
        # dV[s][c] = sum_t P[t][s] gA[t][c]
        self._bgemm(S.data_ptr(), gA.data_ptr(), dbase + vo * fs, T, ch, T, T, C, 3 * C, 1, 1, B_, heads, sS, sA, s3)
        # dP[t][s] = sum_c gA[t][c] V[s][c]
        dP = torch.empty_like(S)
        self._bgemm(gA.data_ptr(), base + vo * fs, dP.data_ptr(), T, T, ch, C, 3 * C, T, 0, 0, B_, heads, sA, s3, sS)
        _lib.check(self.lib.fh_softmax_bwd_rows(S.data_ptr(), dP.data_ptr(), B_ * T, T, _lib.stream()), "softmax_bwd")
        alpha = 1.0 / math.sqrt(ch)
        # dQ[t][c] = alpha sum_s dS[t][s] K[s][c] ;  dK[s][c] = alpha sum_t dS[t][s] Q[t][c]
        self._bgemm(dP.data_ptr(), base + ko * fs, dbase + qo * fs, T, ch, T, T, 3 * C, 3 * C, 0, 1, B_, heads, sS, s3, s3,
                    alpha)
        self._bgemm(dP.data_ptr(), base + qo * fs, dbase + ko * fs, T, ch, T, T, 3 * C, 3 * C, 1, 1, B_, heads, sS, s3, s3,
                    alpha)
        g_hn = self._dgrad(p + ".qkv", dqkv, gn=(p + ".norm", x, st, 0, None, None))
        return self._gn_bwd(p + ".norm", x, st, g_hn, 0, accumulate_into=g, add2=add2)

    # ---------------------------------------------------------------- whole network
    def forward_tape(self, steps, x_nchw, emb):
        cfg = self.cfg
        N, Cin, H, W = x_nchw.shape
        st = _lib.stream()
        self.lib.fh_unet_set_precision(int(self.bf16))  # process-wide switch of the bf16-MFMA convolution kernels
        x = torch.empty(N, H, W, _pad_to(Cin, _K), dtype=torch.float32, device=x_nchw.device)
        _lib.check(self.lib.fh_layout_nchw_nhwc(x_nchw.contiguous().data_ptr(), x.data_ptr(), N, Cin, H * W, x.shape[-1],
                                                1, st), "layout")
        tape, stack, h = [], [], x
        for op, p, _ci, _co, heads in steps:
            if op == "push":
                stack.append(h)
                tape.append(("push",))
            elif op == "pop_cat":
                s = stack.pop()
                Nn, Hh, Ww, Ca = h.shape
                Cb = s.shape[-1]
                out = torch.empty(Nn, Hh, Ww, Ca + Cb, dtype=torch.float32, device=h.device)
                _lib.check(self.lib.fh_concat_channels(h.data_ptr(), s.data_ptr(), out.data_ptr(), Nn * Hh * Ww, Ca, Cb,
                                                       0, st), "concat")
                tape.append(("pop_cat", Ca, Cb))
                h = out
            elif op == "conv_in":
                h = self._conv(p, h)
                tape.append(("conv", p))
            elif op in ("res", "res_down", "res_up"):
                h = self._res_fwd(op, p, h, emb, tape)
            elif op == "attn":
                h = self._attn_fwd(p, h, heads, tape)
            elif op == "down":  # Downsample, openai_unet.py:114-139 (the public 256 x 256 checkpoints use res_down instead)
                if cfg.conv_resample:
                    h = self._conv(p + ".op", h, stride=2)
                    tape.append(("down_conv", p + ".op"))
                else:
                    h = self._resample(h, 0)
                    tape.append(("down_pool",))
            elif op == "up":  # Upsample, openai_unet.py:77-111: nearest x2, then the 3x3 convolution
                h = self._resample(h, 2)
                if cfg.conv_resample:
                    h = self._conv(p + ".conv", h)
                tape.append(("up", p + ".conv" if cfg.conv_resample else None))
            else:
                raise NotImplementedError(f"UNet step '{op}'")
        hn, stn = self._gn("out.0", h, act=1)
        tape.append(("out", h, stn))
        y = self._conv("out.2", hn)  # [N,H,W,out_channels]
        co = y.shape[-1]
        out = torch.empty(N, co, H, W, dtype=torch.float32, device=y.device)
        _lib.check(self.lib.fh_layout_nchw_nhwc(y.data_ptr(), out.data_ptr(), N, co, H * W, co, 0, st), "layout")
        return out, tape

    def backward_tape(self, tape, g_nchw):
        N, Co, H, W = g_nchw.shape
        st = _lib.stream()
        self.lib.fh_unet_set_precision(int(self.bf16))
        self._tls.amax_pool = None  # (slots are written by atomicMax: a fresh zeroed pool per pass)
        cp = self.conv["out.2"].co_p
        g = torch.empty(N, H, W, cp, dtype=torch.float32, device=g_nchw.device)
        _lib.check(self.lib.fh_layout_nchw_nhwc(g_nchw.contiguous().data_ptr(), g.data_ptr(), N, Co, H * W, cp, 1, st),
                   "layout")
        pending = []  # gradients of skip tensors, to be added when their `push` is reached
        recs = list(reversed(tape))
        presplit = None   # (ga, gb) when the block before a pop_cat already wrote the concat's gradient as two tensors
        skip_push = False
        fold = os.environ.get("FH_GLUE_FOLD", "1") != "0"
        for ri, rec in enumerate(recs):
            kind = rec[0]
            nxt = recs[ri + 1] if ri + 1 < len(recs) else (None,)
            if kind == "out":
                _, h, stn = rec
                g = self._dgrad("out.2", g)
                g = self._gn_bwd("out.0", h, stn, g, 1)
            elif kind in ("res", "attn"):
                # the last pass of a block's backward (GroupNorm backward of its first norm) also does the glue around it:
                # + the skip gradient when the block's input was forked by a `push`, or the split of a concat's gradient
                add2 = pending.pop() if (fold and nxt[0] == "push") else None
                skip_push = add2 is not None
                if kind == "res":
                    split = (nxt[1], nxt[2]) if (fold and nxt[0] == "pop_cat" and nxt[1] % 4 == 0) else None
                    g = self._res_bwd(rec, g, add2=add2, split=split)
                    if split is not None:
                        presplit, g = g, None
                else:
                    g = self._attn_bwd(rec, g, add2=add2)
            elif kind == "pop_cat":
                _, Ca, Cb = rec
                if presplit is not None:
                    ga, gb = presplit
                    presplit = None
                else:
                    Nn, Hh, Ww, _ = g.shape
                    ga = torch.empty(Nn, Hh, Ww, Ca, dtype=torch.float32, device=g.device)
                    gb = torch.empty(Nn, Hh, Ww, Cb, dtype=torch.float32, device=g.device)
                    _lib.check(self.lib.fh_concat_channels(ga.data_ptr(), gb.data_ptr(), g.data_ptr(), Nn * Hh * Ww, Ca, Cb,
                                                           1, st), "split")
                pending.append(gb)
                g = ga
            elif kind == "push":
                if skip_push:
                    skip_push = False
                else:
                    g = self._add(g, pending.pop())
            elif kind == "down_conv":  # stride-2 convolution: zero-inserted cotangent through the stride-1 input-gradient pass
                g = self._dgrad(rec[1], self._resample(g, 4))
            elif kind == "down_pool":
                g = self._resample(g, 1)
            elif kind == "up":
                if rec[1] is not None:
                    g = self._dgrad(rec[1], g)
                g = self._resample(g, 3)
            elif kind == "conv":
                g = self._dgrad(rec[1], g)  # [N,H,W,3]
        ci = g.shape[-1]
        out = torch.empty(N, ci, H, W, dtype=torch.float32, device=g.device)
        _lib.check(self.lib.fh_layout_nchw_nhwc(g.data_ptr(), out.data_ptr(), N, ci, H * W, ci, 0, st), "layout")
        return out

    def run(self, steps, x, emb, emb_key=None):
        self._tls.emb_key = emb_key
        return _UNetFn.apply(x, self, steps, emb)


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ops, steps, emb):
        out, tape = ops.forward_tape(steps, x.detach(), emb.detach())
        ctx.ops, ctx.tape = ops, tape
        return out

    @staticmethod
    def backward(ctx, g):
        gx = ctx.ops.backward_tape(ctx.tape, g)
        ctx.tape = None
        return gx, None, None, None
