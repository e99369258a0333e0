"""GroupNorm passes at the UNet's large shapes: time and effective bandwidth of fh_groupnorm_stats / _apply / _bwd"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from free_hunch_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")

def timed(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

for (N, HW, Cc) in ((8, 256 * 256, 128), (8, 256 * 256, 256), (8, 128 * 128, 256), (8, 64 * 64, 512)):
    x = torch.randn(N, HW, Cc, device=dev)
    dy = torch.randn(N, HW, Cc, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    gamma, beta = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    stats = torch.empty(N, 32, 2, device=dev); sums = torch.empty(N, 32, 2, device=dev)
    scratch = torch.empty(lib.fh_groupnorm_scratch_doubles(N, HW), dtype=torch.float64, device=dev)
    st = _lib.stream()
    tb = x.numel() * 4
    t = timed(lambda: lib.fh_groupnorm_stats(x.data_ptr(), stats.data_ptr(), scratch.data_ptr(), N, HW, Cc, st))
    print(f"N={N} P={HW} C={Cc} ({tb/2**20:.0f} MiB): stats {t*1e6:7.1f} us {tb/t/1e12:5.2f} TB/s", end="")
    t = timed(lambda: lib.fh_groupnorm_apply(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, 0, y.data_ptr(), N, HW, Cc, 1, st))
    print(f" | apply {t*1e6:7.1f} us {2*tb/t/1e12:5.2f} TB/s", end="")
    t = timed(lambda: lib.fh_groupnorm_bwd(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, 0, sums.data_ptr(), scratch.data_ptr(), dx.data_ptr(), N, HW, Cc, 1, 0, st))
    print(f" | bwd (partial + stream) {t*1e6:7.1f} us {5*tb/t/1e12:5.2f} TB/s")
