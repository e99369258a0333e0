import os, sys, torch, time, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from free_hunch_amd import _lib, covariance as hc
from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2
from test_hip_parity import _hip_op
dev=torch.device('cuda:0')
S, d = 256, 3*256*256
cov = hc.CovarianceHessianBFGSDCT(os.path.join(ROOT,"free-hunch_amd","data"), 80.0**2, d, device=dev, use_precalculated_info=True)
steps = inputs.script(4242, (1, 3, S, S), 8, 80.0)
for what, a in steps:
    if what == "time": cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
    else: cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
side = torch.cuda.Stream()
for name in ("gaussian_blur","motion_blur","inpainting"):
    op = _hip_op(name, S, dev)
    prob, keep = _problem(op, cov, _sigma_y2(op))
    b = torch.randn(d, dtype=torch.float64, device=dev); sol = torch.empty_like(b); info=_lib.FhCgInfo(); ctx=cov.ctx
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for rep in range(3):
            torch.cuda.synchronize(); t0=time.time()
            _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b.data_ptr(), sol.data_ptr(), 1e-30, 0.0, 64, C.byref(info), _lib.stream()), "cg")
            torch.cuda.synchronize(); dt=time.time()-t0
        print(name, "m", cov.famC.m, "iters", info.niter, "%.1f us/iter" % (dt*1e6/max(1,info.niter)), "graphs_disabled?", os.environ.get("FH_NO_GRAPH"))
