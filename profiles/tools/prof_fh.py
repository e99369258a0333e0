import os, sys, torch, tempfile, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from bench import roofline_cov_apply
from free_hunch_amd import _lib, covariance as hc
from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2
from test_hip_parity import _hip_op
dev=torch.device('cuda:0')
print(roofline_cov_apply(dev, iters=50))
S, d = 256, 3*256*256
cov = hc.CovarianceHessianBFGSDCT(os.path.join(ROOT,"free-hunch_amd","data"), 80.0**2, d, device=dev, use_precalculated_info=True)
steps = inputs.script(4242, (1, 3, S, S), 16, 80.0)
import time
torch.cuda.synchronize(); t0=time.time()
for what, a in steps:
    if what == "time":
        cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
    else:
        cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
torch.cuda.synchronize(); print("16 time+space updates: %.1f ms" % ((time.time()-t0)*1e3))
for name in ("gaussian_blur","motion_blur","inpainting","super_resolution"):
    op = _hip_op(name, S, dev)
    prob, keep = _problem(op, cov, _sigma_y2(op))
    n = d if name!="super_resolution" else d//16
    b = torch.randn(n, dtype=torch.float64, device=dev)
    sol = torch.empty_like(b); info=_lib.FhCgInfo(); ctx=cov.ctx
    _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b.data_ptr(), sol.data_ptr(), 1e-6, 0.0, 8, C.byref(info), _lib.stream()), "cg")
    torch.cuda.synchronize(); t0=time.time()
    _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b.data_ptr(), sol.data_ptr(), 1e-6, 0.0, 64, C.byref(info), _lib.stream()), "cg")
    torch.cuda.synchronize(); dt=time.time()-t0
    print(name, "cg iters", info.niter, "%.1f us/iter" % (dt*1e6/max(1,info.niter)))

# UNet (PyTorch-ROCm bring-up backend) forward + input-VJP at the two benchmark architectures
from bench import build_net
import sys
for arch in ("ffhq", "imagenet"):
  for backend in ("hip",):
    net, cfg = build_net(arch, dev, backend)
    for bs in (1, 8):
      x = torch.randn(bs,3,256,256, device=dev, dtype=torch.float64)
      sig = torch.tensor(5.0, dtype=torch.float64, device=dev)
      for it in range(3):
        torch.cuda.synchronize(); t0=time.time()
        xt = x.clone().requires_grad_()
        D,_ = net(xt, sig)
        torch.cuda.synchronize(); t1=time.time()
        g, = torch.autograd.grad((D*D.detach()).sum(), xt)
        torch.cuda.synchronize(); t2=time.time()
      print(arch, backend, "batch", bs, "UNet fwd %.1f ms  vjp %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3), flush=True)
