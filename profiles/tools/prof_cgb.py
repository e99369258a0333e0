import os, sys, torch, time, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from free_hunch_amd import _lib, covariance as hc
from free_hunch_amd.conditioning_mechanisms import _problem, _sigma_y2
from test_hip_parity import _hip_op
dev=torch.device('cuda:0'); S, d = 256, 3*256*256; Bn=int(sys.argv[1]) if len(sys.argv)>1 else 8; nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 8
covs=[]
for b in range(Bn):
    cov = hc.CovarianceHessianBFGSDCT(os.path.join(ROOT,"free-hunch_amd","data"), 80.0**2, d, device=dev, use_precalculated_info=True, ctx_slot=b)
    for what, a in inputs.script(4242+b, (1, 3, S, S), nsteps, 80.0):
        if what == "time": cov.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
        else: cov.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    covs.append(cov)
side = torch.cuda.Stream()
ctx = _lib.Context.get(S, 3*Bn, 0, slot=5000+Bn)
for name in ("gaussian_blur","inpainting"):
    op = _hip_op(name, S, dev)
    prob, keep = _problem(op, covs[0], _sigma_y2(op))
    per=_lib.FhBatch(); per.nimg=Bn
    for b,cov in enumerate(covs):
        per.D[b],per.r[b],per.B[b],per.M[b]=cov.C.D.data_ptr(),cov.C.r.data_ptr(),cov.famC.B.data_ptr(),cov.C.M_dev.data_ptr()
        if name=="inpainting":
            mk=op.mask.to(device=dev,dtype=torch.float64).contiguous(); keep.append(mk); per.mask[b]=mk.data_ptr()
    bvec = torch.randn(Bn, d, dtype=torch.float64, device=dev); sol = torch.empty_like(bvec)
    infos=(_lib.FhCgInfo*Bn)(); rt=(C.c_double*Bn)(*([1e-30]*Bn))
    with torch.cuda.stream(side):
        for rep in range(3):
            torch.cuda.synchronize(); t0=time.time()
            _lib.check(ctx.lib.fh_cg_solve_batched(ctx.h, C.byref(prob), C.byref(per), bvec.data_ptr(), sol.data_ptr(), rt, 0.0, 64, infos, _lib.stream()), "cg")
            torch.cuda.synchronize(); dt=time.time()-t0
    print(name, "B", Bn, "m", covs[0].famC.m, "iters", [i.niter for i in infos][:3], "%.1f us per batched iteration, %.1f us per image-iteration" % (dt*1e6/64, dt*1e6/64/Bn))
