import os, sys, torch, numpy as np, tempfile, pathlib
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from test_oracle_golden import baseline_inputs
from test_hip_parity import _base_kwargs, _hip_net, _hip_op
from free_hunch_amd.sampler import conditional_sampler
from free_hunch_amd import conditioning_mechanisms as cm
dev=torch.device('cuda:0'); g=np.load(os.path.join(ROOT,'tests','golden','baselines.npz'))
tmp=pathlib.Path(tempfile.mkdtemp())
for tag in sys.argv[1:]:
    c=baseline_inputs(g, tag); net=_hip_net(g, dev, "hip")
    op=_hip_op(c["opname"],64,dev,None)
    sums=[]
    cls=cm.choose_conditioning_mechanism(c["mech"])
    orig=cls.x0_mean_update
    def rec(self,x_t,model,y,sigma,orig=orig):
        out=orig(self,x_t,model,y,sigma); sums.append(float(out.detach().double().sum())); return out
    cls.x0_mean_update=rec
    kw=_base_kwargs(tmp, {"conditioning_mechanism": c["mech"], "diffpir_lambda": 10.0, "pigdm_posthoc_scaling": False, **c["over"]})
    x,_,_=conditional_sampler(net, c["noise"].to(dev), None, None, num_steps=c["nsteps"], sigma_min=0.002, sigma_max=80, rho=7, solver=c["solver"], measurement=c["y"].to(dev), operator=op, **kw)
    cls.x0_mean_update=orig
    ref=torch.from_numpy(g[c["p"]+"x_final"])
    print(tag, "max err", float((x.detach().cpu()-ref).abs().max()))
    rs=g[c["p"]+"out_sum"]
    for i,(a,b) in enumerate(zip(sums, rs)): print("  call",i,"%.6f %.6f diff %.2e"%(a,b,a-b))
