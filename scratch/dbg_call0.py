import os, sys, numpy as np, torch, tempfile
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from test_hip_parity import _hip_net, _hip_op, T, maxabs
from test_oracle_golden import _mk_op
from oracle import fh_oracle as fo, unet_oracle as uo
from free_hunch_amd import covariance as hc
from free_hunch_amd.conditioning_mechanisms import choose_solver
g=np.load(os.path.join(ROOT,'tests/golden/trajectories.npz'))
dev=torch.device('cuda:0')
tmp=tempfile.mkdtemp()
torch.save(T(g["dct_variance64"]), os.path.join(tmp,"dct_variance.pt"))
tag=sys.argv[1]; p=tag+"__"
opname=str(g[p+"op"])
s_img, s_noise = (int(v) for v in g[p + "seeds"])
cfg=inputs.SMALL_A
onet = fo.LinearPrecond(uo.OracleUNet(cfg, uo.seeded_state(cfg, 11)))
hnet = _hip_net(g, dev, "torch")
mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
hop = _hip_op(opname, 64, dev, mask)
oop = _mk_op(opname, 64, g, p)
x0img = inputs.smooth_image(64, s_img)
if opname!="inpainting": oop.forward(x0img.clone())
noise = inputs.randn((1, 3, 64, 64), s_noise, torch.float32)
y = T(g[p + "y"])
t = fo.edm_sigma_steps(onet.u, int(g[p+"num_steps"]))
sig = t[0]
x_t = noise.to(torch.float64)*sig
with torch.no_grad():
    m_o,_ = onet(x_t, sig)
    m_h,_ = hnet(x_t.to(dev), sig.to(dev))
print("unet mean diff", maxabs(m_o, m_h), "absmax", float(m_o.abs().max()))
d=3*64*64
ocov = fo.make_covariance("dct_diagonal", tmp, float(sig**2), d)
hcov = hc.CovarianceHessianBFGSDCT(tmp, float(sig**2), d, device=dev, use_precalculated_info=True)
probe = inputs.randn((1,3,64,64), 3)
print("cov apply diff", maxabs(ocov.denoiser_cov_vector_dot(probe), hcov.denoiser_cov_vector_dot(probe.to(dev))))
io=[]; ih=[]
mat_o = fo.solve_mat(oop, y, m_o, ocov, 1.0, float(sig), io)
mat_h = choose_solver(opname, hop, y.to(dev), m_o.to(dev), None, hcov, "customcuda", 1.0, sigma_t=float(sig), info_out=ih)
print("info", io, ih)
print("mat diff", maxabs(mat_o, mat_h), "absmax", float(mat_o.abs().max()))
# tighter tolerance solve
for rt in (1e-2, 1e-6):
    import free_hunch_amd.conditioning_mechanisms as cmh
    so,info_o = None, []
    # oracle with custom rtol
    orig = fo.rtol_func; fo.rtol_func = lambda s, m=1.0: rt
    cmh_orig = cmh.rtol_func; cmh.rtol_func = lambda s, m=1.0: rt
    a=[];b=[]
    mo = fo.solve_mat(oop, y, m_o, ocov, 1.0, float(sig), a)
    mh = choose_solver(opname, hop, y.to(dev), m_o.to(dev), None, hcov, "customcuda", 1.0, sigma_t=float(sig), info_out=b)
    fo.rtol_func=orig; cmh.rtol_func=cmh_orig
    print("rtol",rt,"niter",a[0]["niter"],b[0]["niter"],"mat diff",maxabs(mo,mh),"absmax",float(mo.abs().max()))
