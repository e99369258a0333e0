"""The UNet's dominant convolution (3x3, 128 -> 128 on 8 x 256 x 256 NHWC; also 256 -> 256 at 128^2 and 512 -> 512 at 64^2) in the
exact bf16-split mode (0) and the half-split mode (4): microseconds and fp32-equivalent TFLOP/s, plain and fused-GroupNorm
input.  `python3 profiles/tools/bench_conv_modes.py [mode ...]`"""
import ctypes as C, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from free_hunch_amd import _lib
from free_hunch_amd.unet_hip import _split3, _half_split_planes
lib = _lib.load(); dev = torch.device("cuda:0")
modes = [int(m) for m in sys.argv[1:]] or [0, 4]
iters = int(os.environ.get("ITERS", "20"))
res = {}
for (N, H, W, Ci, Co) in ((8, 256, 256, 128, 128), (8, 128, 128, 256, 256), (8, 64, 64, 512, 512)):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, H, W, Ci, generator=g).to(dev)
    w = (torch.randn(Co, 9, Ci, generator=g) * 0.03).to(dev)
    b = torch.zeros(Co, device=dev); out = torch.empty(N, H, W, Co, device=dev)
    table = torch.stack([torch.ones(N, Ci), torch.zeros(N, Ci)], 1).contiguous().to(dev)
    amax = torch.empty(16, device=dev)
    _lib.check(lib.fh_absmax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), _lib.stream()), "amax")
    e = _lib.FhGnEpilogue(); e.partial, e.in_amax = None, amax.data_ptr()
    for mode in modes:
        planes = _split3(w) if mode == 0 else _half_split_planes(w)
        lib.fh_unet_set_precision(mode)
        plain = lambda: _lib.check(lib.fh_conv2d_x6_nhwc_gn(x.data_ptr(), planes.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, 1,
                                                            N, H, W, Ci, Co, 3, 3, 1, 1, C.byref(e), _lib.stream()), "conv")
        fused = lambda: _lib.check(lib.fh_conv2d_x6_norm_nhwc(x.data_ptr(), table.data_ptr(), 1, planes.data_ptr(), b.data_ptr(), None,
                                                              out.data_ptr(), N, H, W, Ci, Co, _lib.stream()), "fused")
        for name, f in (("plain", plain), ("gn_fused", fused)):
            for _ in range(3):
                f()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(iters):
                f()
            t1.record(); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) / iters * 1e3
            res[f"{H}x{W}_{Ci}->{Co}_mode{mode}_{name}"] = dict(us=round(us, 1), tflops_fp32_equiv=round(2.0 * N * H * W * Ci * Co * 9 / us / 1e6, 1))
    lib.fh_unet_set_precision(0)
print(json.dumps(res, indent=1))
