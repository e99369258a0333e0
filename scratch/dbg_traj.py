import os, sys, numpy as np, torch
os.environ['FH_TRACE_SUMS']='1'
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
import inputs, tempfile
from test_hip_parity import _hip_net, _hip_op, T, maxabs
from free_hunch_amd.sampler import conditional_sampler
g=np.load(os.path.join(ROOT,'tests/golden/trajectories.npz'))
dev=torch.device('cuda:0')
tmp=tempfile.mkdtemp()
torch.save(T(g["dct_variance64"]), os.path.join(tmp,"dct_variance.pt"))
for tag in sys.argv[1:]:
    p=tag+"__"
    over = eval(str(g[p + "over"]))
    opname, solver, nsteps = str(g[p + "op"]), str(g[p + "solver"]), int(g[p + "num_steps"])
    s_img, s_noise = (int(v) for v in g[p + "seeds"])
    if os.environ.get("FH_CPU_NET"):
        from oracle import fh_oracle as fo, unet_oracle as uo
        onet = fo.LinearPrecond(uo.OracleUNet(inputs.SMALL_A, uo.seeded_state(inputs.SMALL_A, 11)))
        class W:
            sigma_min, sigma_max, u = onet.sigma_min, onet.sigma_max, onet.u
            def round_sigma(self, s): return onet.round_sigma(torch.as_tensor(s).cpu()).to(dev)
            def __call__(self, x, s):
                a, b = onet(x.cpu(), torch.as_tensor(s).cpu())
                return a.to(dev), b.to(dev)
        net = W()
    else:
        net = _hip_net(g, dev, os.environ.get("FH_UNET_BACKEND", "torch"))
    mask = T(g[p + "mask"]).float().repeat(1, 3, 1, 1) if opname == "inpainting" else None
    op = _hip_op(opname, 64, dev, mask)
    noise = inputs.randn((1, 3, 64, 64), s_noise, torch.float32).to(dev)
    y = T(g[p + "y"]).to(dev)
    base = dict(conditioning_mechanism="online_covariance", cond_scaling=1.0, clip_x0_mean=False,
                max_vector_count=100000, dataset_path=tmp, image_base_covariance="dct_diagonal",
                denoiser_mean_error_threshold=0.2, use_analytical_score_time_update=True, project_to_diagonal=False,
                space_step_update_threshold=10.0, space_step_update_lower_threshold=1.0, max_rtol=1.0,
                do_space_updates=True)
    x, _, _ = conditional_sampler(net, noise, None, None, num_steps=nsteps, sigma_min=0.002, sigma_max=80, rho=7,
                                  solver=solver, measurement=y, operator=op, **{**base, **over})
    tr = conditional_sampler.last_mechanism.trace
    print(tag)
    print(" sigma diff", [float(a["sigma"]-b) for a,b in zip(tr, g[p+"sigma"]) if a["sigma"]!=b][:5])
    print(" k   ", [t["k"] for t in tr] == list(g[p + "k"]))
    print(" br  ", [int(t["branch"] == "cov") for t in tr], list(g[p + "branch_cov"]))
    print(" nit ", [t["niter"] for t in tr]); print(" ref ", list(g[p + "niter"]))
    print(" out_sum diff", ["%.2e" % (t["out_sum"]-b) for t,b in zip(tr, g[p+"out_sum"])])
    print(" out_sum ref ", ["%.3f" % b for b in g[p+"out_sum"]])
    print(" xfinal ref absmax", float(np.abs(g[p+"x_final"]).max()))
    print(" final maxabs", maxabs(x, g[p + "x_final"]))
