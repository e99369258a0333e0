import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/tests/golden"]
os.chdir(ROOT)
import inputs
import test_hip_parity as tp
from oracle import fh_oracle as fo
torch.set_num_threads(16)
dev = torch.device("cuda:0")
import tempfile
tmp = tempfile.mkdtemp()
g = np.load("tests/golden/solver.npz", allow_pickle=False)
torch.save(tp.T(g["dct_variance64"]), tmp + "/dct_variance.pt")
meta = tp.CASES[6]
orc, hip = tp._mk_pair(meta, tmp, dev)
steps = inputs.script(906, meta["shape"], meta["n"], meta["sigma0"], meta["neg"])
z = inputs.randn(meta["shape"], 98, torch.float64).reshape(-1)
def cons():
    zo = z.to(torch.complex128)
    t_o = fo._apply_rep(orc.icov, zo)
    co = float((fo._apply_rep(orc.cov, t_o).real - z).abs().max())
    zh = z.to(dev)
    t1 = hip._apply(hip.Ci, hip.famC, zh, torch.empty_like(zh))
    ch = float((hip._apply(hip.C, hip.famC, t1, torch.empty_like(zh)).cpu() - z).abs().max())
    t3 = hip._apply(hip.Hi, hip.famH, zh, torch.empty_like(zh))
    hh = float((hip._apply(hip.H, hip.famH, t3, torch.empty_like(zh)).cpu() - z).abs().max())
    icd = float((t1.cpu() - t_o.real).abs().max() / t_o.real.abs().max())
    return dict(cons_o=co, cons_h=ch, cons_hess_h=hh, icovdiff=icd)
print("init", cons())
for si, (what, a) in enumerate(steps):
    if what == "time":
        orc.update_time_step(a["x"], a["sigma"], a["sigma_next"], a["score"])
        hip.update_time_step(a["x"].to(dev), a["sigma"], a["sigma_next"], a["score"].to(dev))
    else:
        orc.update_space_step(a["m0"], a["m1"], a["sigma"], a["x"], a["xn"])
        hip.update_space_step(a["m0"].to(dev), a["m1"].to(dev), a["sigma"], a["x"].to(dev), a["xn"].to(dev))
    print(si, what, a["sigma"], hip.k, {k: float(f"{v:.3g}") for k, v in cons().items()}, flush=True)
    if si > 14: break
