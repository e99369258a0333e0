"""Phase timeline of the single-sweep cov-apply (k_rep_fused): FH_FUSED_DEBUG=1 makes thread 0 of every workgroup record the
100 MHz wall clock at the phase boundaries.  Prints, per image, when (us after the first workgroup started) the first /
median / last workgroup passed each boundary."""
import os, sys, ctypes as C
os.environ["FH_FUSED_DEBUG"] = "1"
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from free_hunch_amd import _lib
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
ctx.set_exclusive(2)
f = lambda: _lib.check(ctx.lib.fh_rep_apply_batched(ctx.h, C.byref(per), 64, z.data_ptr(), out.data_ptr(), d, m, _lib.stream()), "apply")
for _ in range(5): f()
torch.cuda.synchronize()
nb = d // 768
buf = (C.c_ulonglong * (nimg * nb * 8))()
_lib.check(ctx.lib.fh_debug_read_stamps(ctx.h, buf, nimg * nb * 8, _lib.stream()), "stamps")
t = np.frombuffer(buf, dtype=np.uint64).reshape(nimg, nb, 8).astype(np.int64)
t0 = t[:, :, 0].min()
names = ["start", "loads+dots issued", "stores drained+barrier", "poll passed", "partials read", "chains in LDS", "end"]
for i in range(nimg):
    print(f"image {i}")
    for k, nm in enumerate(names):
        v = (t[i, :, k] - t0) / 100.0
        print(f"   {nm:26s} first {v.min():7.2f}  median {np.median(v):7.2f}  last {v.max():7.2f} us")
