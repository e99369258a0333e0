"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the CPU oracle.

A 64x64 super-resolution Free Hunch run (Heun, 10 steps = 19 guidance calls: UNet forward + input-VJP, covariance
time/space updates, CG solves through the operator) with the product on the GPU, against the oracle on the host."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = torch.device("cuda:0")
    from oracle import fh_oracle as fo, unet_oracle as uo  # the checker
    from free_hunch_amd import unet as hu
    from free_hunch_amd.measurements import get_operator
    from free_hunch_amd.precond import iDDPMLinearPrecond
    from free_hunch_amd.sampler import conditional_sampler
    import scipy.io

    S = 64
    tmp = tempfile.mkdtemp()
    dv = torch.load(os.path.join(ROOT, "free-hunch_amd", "data", "dct_variance.pt"), weights_only=True)
    torch.save(dv[:, :S, :S].contiguous(), os.path.join(tmp, "dct_variance.pt"))
    kw = dict(image_size=S, num_channels=32, num_res_blocks=1, channel_mult=(), learn_sigma=True,
              attention_resolutions="16,8", num_heads=4, num_head_channels=32, use_scale_shift_norm=True,
              resblock_updown=True, use_new_attention_order=False)
    ocfg, hcfg = uo.UNetConfig(**kw), hu.UNetConfig(**kw)
    sd = uo.seeded_state(ocfg, 11)
    onet = fo.LinearPrecond(uo.OracleUNet(ocfg, sd))
    model = hu.UNetModel(hcfg, backend=os.environ.get("FH_UNET_BACKEND", "hip"))
    model.load_state_dict(hu.seeded_state(hcfg, 11))
    hnet = iDDPMLinearPrecond(model.to(dev).eval(), S, 3).to(dev)

    g = torch.Generator().manual_seed(3)
    x0 = torch.tanh(torch.nn.functional.avg_pool2d(torch.randn(1, 3, S * 4, S * 4, generator=g), 4) * 2)
    noise = torch.randn(1, 3, S, S, generator=g)
    kernel = scipy.io.loadmat(os.path.join(ROOT, "free-hunch_amd", "data", "kernels", "kernels_bicubicx234.mat"))[
        "kernels"][0, 2].astype(np.float64)
    oop = fo.OracleOperator("super_resolution", (1, 3, S, S), 0.1, kernel=kernel, scale_factor=4)
    y = oop.forward(x0, noise=torch.randn(1, 3, S // 4, S // 4, generator=g))
    hop = get_operator(name="super_resolution", device=dev, sigma_s=0.1, scale_factor=4, in_shape=(1, 3, S, S))

    fac = lambda op_, v0, d: fo.OracleFreeHunch(1.0, op_, False, v0, d, image_base_covariance="dct_diagonal",
                                                data_dir=tmp)
    xo, mo = fo.conditional_sampler(onet, noise, y, oop, num_steps=10, solver="heun", mechanism_factory=fac)
    xh, _, _ = conditional_sampler(
        hnet, noise.to(dev), None, None, num_steps=10, sigma_min=0.002, sigma_max=80, rho=7, solver="heun",
        measurement=y.to(dev), operator=hop, conditioning_mechanism="online_covariance", cond_scaling=1.0,
        clip_x0_mean=False, max_vector_count=100000, dataset_path=tmp, image_base_covariance="dct_diagonal",
        denoiser_mean_error_threshold=0.2, use_analytical_score_time_update=True, project_to_diagonal=False,
        space_step_update_threshold=10.0, space_step_update_lower_threshold=1.0, max_rtol=1.0, do_space_updates=True)
    mh = conditional_sampler.last_mechanism
    err = float((xh.cpu() - xo).abs().max())
    print(f"smoke: 64x64 SR Heun-10, {len(mh.trace)} guidance calls, CG iters {[t['niter'] for t in mh.trace]} "
          f"(oracle {[t['niter'] for t in mo.trace]}), k={mh.trace[-1]['k']}, max|x_hip - x_oracle| = {err:.3e}")
    assert [t["k"] for t in mh.trace] == [t["k"] for t in mo.trace]
    assert err < 1e-3, err
