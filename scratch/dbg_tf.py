"""teacher-forced per-call comparison: the oracle drives the trajectory; the HIP plugin sees the same (x_t, x0_mean)."""
import os, sys, numpy as np, torch, tempfile
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs
from test_hip_parity import _hip_op, T, maxabs
from test_oracle_golden import _mk_op
from oracle import fh_oracle as fo, unet_oracle as uo
from free_hunch_amd.conditioning_mechanisms import BFGSOnlineUpdate
g=np.load(os.path.join(ROOT,'tests/golden/trajectories.npz'))
dev=torch.device('cuda:0')
tmp=tempfile.mkdtemp()
torch.save(T(g["dct_variance64"]), os.path.join(tmp,"dct_variance.pt"))
cfg=inputs.SMALL_A
onet = fo.LinearPrecond(uo.OracleUNet(cfg, uo.seeded_state(cfg, 11)))
for tag in sys.argv[1:]:
    p=tag+"__"; opname=str(g[p+"op"]); over=eval(str(g[p+"over"]))
    s_img, s_noise = (int(v) for v in g[p + "seeds"])
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
    hop = _hip_op(opname, 64, dev, mask); oop = _mk_op(opname, 64, g, p)
    if opname!="inpainting": oop.forward(inputs.smooth_image(64, s_img))
    noise = inputs.randn((1, 3, 64, 64), s_noise, torch.float32); y = T(g[p + "y"])
    kw = dict(image_base_covariance=over.get("image_base_covariance", "dct_diagonal"), data_dir=tmp,
              do_space_updates=over.get("do_space_updates", True),
              space_step_update_threshold=over.get("space_step_update_threshold", 10.0),
              space_step_update_lower_threshold=over.get("space_step_update_lower_threshold", 1.0))
    state={}
    class Pair:
        def __init__(self, op_, v0, d):
            self.o = fo.OracleFreeHunch(1.0, op_, False, v0, d, **kw)
            self.h = BFGSOnlineUpdate(1.0, hop, False, 1, torch.as_tensor(v0), d, max_vector_count=100000, data_dir=tmp,
                     image_base_covariance=kw["image_base_covariance"], denoiser_mean_error_threshold=0.2,
                     use_analytical_score_time_update=True, project_to_diagonal=False,
                     space_step_update_threshold=kw["space_step_update_threshold"],
                     space_step_update_lower_threshold=kw["space_step_update_lower_threshold"], max_rtol=1.0,
                     do_space_updates=kw["do_space_updates"], solver_type="customcuda")
            self.rows=[]
        def __call__(self, x_t, net, y_, sigma):
            out_o = self.o(x_t, net, y_, sigma)
            def net_dev(x, s):
                a, b = net(x.cpu(), torch.as_tensor(s).cpu())
                return a.to(dev), b.to(dev)
            out_h = self.h(x_t.to(dev).clone(), net_dev, y_.to(dev), sigma.to(dev))
            to, th = self.o.trace[-1], self.h.trace[-1]
            self.rows.append((float(sigma), to["niter"], th["niter"], to["branch"], th["branch"], to["k"], th["k"],
                              maxabs(out_o, out_h), float(out_o.abs().max())))
            return out_o
    holder={}
    def fac(op_, v0, d):
        holder["p"]=Pair(op_, v0, d); return holder["p"]
    x, _ = fo.conditional_sampler(onet, noise, y, oop, num_steps=int(g[p+"num_steps"]), solver=str(g[p+"solver"]), mechanism_factory=fac)
    print(tag)
    for r in holder["p"].rows:
        print("  s=%9.4f nit %4d %4d  br %s %s  k %d %d  err %.3e  |out| %.3e  rel %.2e" % (r+(r[7]/r[8],)))
