"""Host-side cost of the covariance updates (enqueue time without synchronising), one thread, 256 x 256, growing factor"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from free_hunch_amd import covariance as hc
dev = torch.device("cuda:0"); S = 256; d = 3 * S * S
cov = hc.CovarianceHessianBFGSDCT(os.path.join(ROOT, "free-hunch_amd", "data"), 80.0 ** 2, d, device=dev, use_precalculated_info=True)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 3, S, S, generator=g, dtype=torch.float64).to(dev)
sc = (torch.randn(1, 3, S, S, generator=g, dtype=torch.float64) * 0.01).to(dev)
sig = 8.0
for j in range(12):
    dx = (torch.randn(1, 3, S, S, generator=g, dtype=torch.float64) * 0.1).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cov.update_time_step(x, sig, sig * 0.9, sc)
    t1 = time.perf_counter()
    cov.update_space_step(x * 0.5, x * 0.5 + 0.4 * dx, sig * 0.9, x, x + dx)
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"m={cov.famC.m:3d} host: time update {1e6*(t1-t0):7.1f} us, space update {1e6*(t2-t1):7.1f} us; device drain {1e6*(t3-t2):7.1f} us")
    x = x + dx; sig *= 0.9
