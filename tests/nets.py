"""Denoisers shared by the GPU tests: the two for which the reference reproduces itself across hosts
(tests/golden/inputs.py: gauss_prior_denoise, damped_state; recorded by make_golden.py as `*_gauss` / `*_damped`)."""
import os

import torch

import inputs


def gauss_net(size, dev):
    """inputs.gauss_prior_denoise behind the product's precond interface (sigma table, round_sigma, sigma_min / sigma_max):
    elementwise float64 torch ops on the device, so its values do not depend on the batch size."""
    from free_hunch_amd.precond import iDDPMLinearPrecond

    class GaussPriorNet(iDDPMLinearPrecond):
        def __init__(self):
            super().__init__(None, size, 3)

        def forward(self, x, sigma, **kw):
            sigma = torch.as_tensor(sigma, device=x.device).to(torch.double).reshape(-1, 1, 1, 1)
            return inputs.gauss_prior_denoise(x, sigma), None

    return GaussPriorNet().to(dev)


def pp_gauss_net(size, dev):
    """inputs.pp_gauss_prior_denoise (per-pixel prior variance inputs.pp_prior_var) behind the product's precond interface."""
    from free_hunch_amd.precond import iDDPMLinearPrecond

    class PerPixelGaussPriorNet(iDDPMLinearPrecond):
        def __init__(self):
            super().__init__(None, size, 3)
            self.register_buffer("var", inputs.pp_prior_var(size))

        def forward(self, x, sigma, **kw):
            sigma = torch.as_tensor(sigma, device=x.device).to(torch.double).reshape(-1, 1, 1, 1)
            return inputs.pp_gauss_prior_denoise(x, sigma, self.var), None

    return PerPixelGaussPriorNet().to(dev)


def _hip_cfg(cfg_in):
    from free_hunch_amd import unet as hu
    return hu.UNetConfig(**{k: getattr(cfg_in, k) for k in
                            ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma",
                             "attention_resolutions", "num_heads", "num_head_channels", "use_scale_shift_norm",
                             "resblock_updown", "use_new_attention_order")})


def damped_hip_net(cfg_in, seed, dev, damp=inputs.DAMP, dtype=None):
    """The HIP UNet with inputs.damped_state weights (the recording side: make_golden.damped_net); `dtype`: UNetModel.set_dtype."""
    from free_hunch_amd import unet as hu
    from free_hunch_amd.precond import iDDPMLinearPrecond
    cfg = _hip_cfg(cfg_in)
    model = hu.UNetModel(cfg, backend=os.environ.get("FH_UNET_BACKEND", "hip"), dtype=dtype or "fp32")
    model.load_state_dict(inputs.damped_state(hu.seeded_state, cfg, seed, damp))
    return iDDPMLinearPrecond(model.to(dev).eval(), cfg.image_size, 3).to(dev)


def oracle_gauss_net():
    """The same closed form behind the oracle's net interface (CPU)."""
    from oracle import fh_oracle as fo
    u = fo.linear_sigma_table()

    class Net:
        sigma_min, sigma_max = float(u[-2]), float(u[0])

        def __init__(self):
            self.u = u

        def round_sigma(self, s):
            return fo.round_sigma(u, s)

        def __call__(self, x, sigma):
            return inputs.gauss_prior_denoise(x, torch.as_tensor(sigma, dtype=torch.float64)), None

    return Net()


def oracle_damped_net(cfg, seed):
    from oracle import fh_oracle as fo, unet_oracle as uo
    return fo.LinearPrecond(uo.OracleUNet(cfg, inputs.damped_state(uo.seeded_state, cfg, seed)))
