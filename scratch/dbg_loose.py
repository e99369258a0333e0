import os, sys, torch, ctypes as C, tempfile, pathlib
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from test_edge_cases import _cov_and_problem, _amm
from free_hunch_amd import _lib
from oracle import fh_oracle as fo
dev=torch.device('cuda:0')
tmp=pathlib.Path(tempfile.mkdtemp())
cov, op, prob, keep = _cov_and_problem(dev, tmp, 64, 0)
ctx=cov.ctx; A=_amm(ctx, prob)
b=inputs.randn((3*64*64,),31).to(dev)
for rtol in (0.9, 0.5, 0.2, 0.05):
    sol=torch.empty_like(b); info=_lib.FhCgInfo()
    _lib.check(ctx.lib.fh_cg_solve(ctx.h, C.byref(prob), b.data_ptr(), sol.data_ptr(), rtol, 0.0, 50, C.byref(info), _lib.stream()),"cg")
    x_ref,i_ref=fo.cg(lambda u: A(u.contiguous()), b, rtol=rtol, atol=0.0, maxiter=50)
    r_h=float((b-A(sol)).norm()/b.norm()); r_o=float((b-A(x_ref.contiguous())).norm()/b.norm())
    print(rtol, "hip niter", info.niter, info.optimal, "res", info.residual_norm/float(b.norm()), "true", r_h, "| oracle", i_ref["niter"], i_ref["optimal"], float(i_ref["residual_norm"]/b.norm()), "true", r_o, "| diff", float((sol-x_ref).abs().max()))
