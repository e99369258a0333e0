"""One mode of the cov-apply at the headline point, for rocprofv3: python3 prof_cov_one.py <nimg> <exclusive 0|1|2> [m]"""
import os, sys, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from free_hunch_amd import _lib
nimg, excl = int(sys.argv[1]), int(sys.argv[2]); m = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0"); S, d = 256, 3 * 256 * 256
ctx = _lib.Context.get(S, 3 * nimg, 256)
g = torch.Generator().manual_seed(1)
Bs = [torch.randn(m, d, generator=g, dtype=torch.float64).to(dev) for _ in range(nimg)]
Ds = [(torch.rand(d, generator=g, dtype=torch.float64) + 0.5).to(dev) for _ in range(nimg)]
rs = [(torch.rand(d, generator=g, dtype=torch.float64) + 0.5).to(dev) for _ in range(nimg)]
Ms = [torch.randn(64, 64, generator=g, dtype=torch.float64).to(dev) for _ in range(nimg)]
z = torch.randn(nimg, d, generator=g, dtype=torch.float64).to(dev); out = torch.empty_like(z)
per = _lib.FhBatch(); per.nimg = nimg
for i in range(nimg):
    per.D[i], per.r[i], per.B[i], per.M[i] = Ds[i].data_ptr(), rs[i].data_ptr(), Bs[i].data_ptr(), Ms[i].data_ptr()
ctx.set_exclusive(excl)
for _ in range(30):
    _lib.check(ctx.lib.fh_rep_apply_batched(ctx.h, C.byref(per), 64, z.data_ptr(), out.data_ptr(), d, m, _lib.stream()), "apply")
torch.cuda.synchronize()
ctx.status()
