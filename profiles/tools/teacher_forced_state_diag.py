"""diagnostic: teacher-forced ip256 Heun-12; for calls with equal iteration counts and rel > 1e-5, re-solve both sides tightly"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/tests/golden"]
os.chdir(ROOT)
import test_hip_parity256 as t
from test_oracle_golden import _mk_op
import inputs
from oracle import fh_oracle as fo, unet_oracle as uo
from free_hunch_amd.conditioning_mechanisms import BFGSOnlineUpdate, solve_customcuda
torch.set_num_threads(32)
dev = torch.device("cuda:0")
g = np.load("tests/golden/trajectories256.npz", allow_pickle=False)
opname, tag = sys.argv[1], sys.argv[2]
size, ncalls, nsteps = 256, 23, 12
p = tag + "__"
s_img, s_noise = (int(v) for v in g[p + "seeds"])
T = t.T
mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
hop, oop = t._hip_op(opname, size, dev, mask), _mk_op(opname, size, g, p)
if opname != "inpainting":
    oop.forward(inputs.smooth_image(size, s_img))
noise, y = inputs.randn((1, 3, size, size), s_noise, torch.float32), T(g[p + "y"])
kw = t._base_kwargs(t.DATA, {})
onet = fo.LinearPrecond(uo.OracleUNet(inputs.SMALL_C, uo.seeded_state(inputs.SMALL_C, int(g["unet_seed"]))))
rows = []
class Stop(Exception): pass
class Pair:
    def __init__(self, op_, v0, d):
        self.o = fo.OracleFreeHunch(1.0, op_, False, v0, d, image_base_covariance="dct_diagonal", data_dir=t.DATA)
        self.h = BFGSOnlineUpdate(1.0, hop, False, 1, torch.as_tensor(v0), d, solver_type="customcuda", data_dir=t.DATA,
                                  **{k: v for k, v in kw.items() if k not in ("conditioning_mechanism", "cond_scaling", "clip_x0_mean", "dataset_path")})
    def __call__(self, x_t, net, y_, sigma):
        out_o = self.o(x_t, net, y_, sigma)
        def net_dev(x, s_):
            a, b = net(x.cpu(), torch.as_tensor(s_).cpu())
            return a.to(dev), b.to(dev)
        out_h = self.h(x_t.to(dev).clone(), net_dev, y_.to(dev), sigma.to(dev))
        to, th = self.o.trace[-1], self.h.trace[-1]
        r = dict(sigma=float(sigma), no=to["niter"], nh=th["niter"], k=to["k"], err=t.maxabs(out_o, out_h), mag=float(out_o.abs().max()), rtol=float(th["rtol"]))
        x0o, x0h = self.o.means[-1], self.h.denoiser_means[-1]
        r["x0diff"] = t.maxabs(x0o, x0h)
        # covariance probes
        v = inputs.randn((1, 3, size, size), 99, torch.float64)
        co, ch = self.o.cov.denoiser_cov_vector_dot(v), self.h.covariance_model.denoiser_cov_vector_dot(v.to(dev))
        r["covdiff"] = t.maxabs(co, ch) / float(co.abs().max())
        # inverse consistency of each side: C (Ci z) = z in the transform domain
        z = inputs.randn((1, 3, size, size), 98, torch.float64).reshape(-1)
        zo = z.to(torch.complex128)
        ro = fo._apply_rep(self.o.cov.cov, fo._apply_rep(self.o.cov.icov, zo)).real
        r["cons_o"] = float((ro - z).abs().max())
        cm = self.h.covariance_model
        zh = z.to(dev)
        t1 = cm._apply(cm.Ci, cm.famC, zh, torch.empty_like(zh))
        rh = cm._apply(cm.C, cm.famC, t1, torch.empty_like(zh))
        r["cons_h"] = float((rh - zh).abs().max())
        # cross: C_o (Ci_h z), C_h (Ci_o z)
        r["cross_Co_Cih"] = float((fo._apply_rep(self.o.cov.cov, t1.cpu().to(torch.complex128)).real - z).abs().max())
        t2 = fo._apply_rep(self.o.cov.icov, zo).real.contiguous().to(dev)
        r["cross_Ch_Cio"] = float((cm._apply(cm.C, cm.famC, t2, torch.empty_like(zh)).cpu() - z).abs().max())
        r["icovdiff"] = float((t2 - t1).abs().max() / t2.abs().max())
        if len(rows) in (18, 20, 21):  # dump the C / C^-1 representations around the last time updates
            cmm = self.h.covariance_model
            m_ = cmm.famC.m
            np.savez(f"/tmp/state_{opname}_{len(rows)}.npz", sigma=float(sigma), m=m_,
                     D=cmm.C.D.cpu().numpy(), r=cmm.C.r.cpu().numpy(), M=cmm.C.M_dev[:m_, :m_].cpu().numpy(),
                     B=cmm.famC.B[:m_].cpu().numpy(), Di=cmm.Ci.D.cpu().numpy(), ri=cmm.Ci.r.cpu().numpy(),
                     Mi=cmm.Ci.M_dev[:m_, :m_].cpu().numpy(),
                     z=z.numpy(), Cz_hip=cmm._apply(cmm.C, cmm.famC, zh, torch.empty_like(zh)).cpu().numpy(),
                     Cz_orc=fo._apply_rep(self.o.cov.cov, zo).real.numpy())
        print({k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
        rows.append(r)
        if len(rows) >= ncalls: raise Stop()
        return out_o
try:
    fo.conditional_sampler(onet, noise, y, oop, num_steps=nsteps, solver="heun", mechanism_factory=lambda op_, v0, d: Pair(op_, v0, d))
except Stop:
    pass
