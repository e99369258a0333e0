"""iDDPM linear-schedule preconditioning (reference: training/openai_preconditioning.py:93-207).
net(x, sigma) -> (D_x, x0_var): D_x = clamp(x - sigma * F(c_in x, M - idx(sigma)), -1, 1) in float64."""
from __future__ import annotations

import numpy as np
import torch


class iDDPMLinearPrecond(torch.nn.Module):
    def __init__(self, model, img_resolution, img_channels, label_dim=0, use_fp16=False, beta_min=0.0001,
                 beta_max=0.02, M=1000, **model_kwargs):
        super().__init__()
        if use_fp16:  # the reference casts the torso and its input to fp16 (:171): half-precision convolution operands
            model.set_dtype("fp16")
        self.use_fp16 = use_fp16
        self.img_resolution, self.img_channels, self.label_dim = img_resolution, img_channels, label_dim
        self.beta_min, self.beta_max, self.M, self.model = beta_min, beta_max, M, model
        betas = torch.cat([torch.tensor([0.0]), torch.linspace(beta_min, beta_max, M)])
        abar = torch.cumprod(1 - betas, dim=0).flip(dims=[0])
        u = torch.sqrt((1 - abar) / abar)
        self.register_buffer("u", u)
        self.sigma_min, self.sigma_max = float(u[M - 1]), float(u[0])
        b = betas.numpy()
        ac = np.cumprod(1.0 - b, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        with np.errstate(divide="ignore", invalid="ignore"):
            pv = b * (1.0 - ac_prev) / (1.0 - ac)
            pc = b * np.sqrt(ac_prev) / (1.0 - ac)
        self.register_buffer("posterior_variance", torch.from_numpy(pv).float())
        self.register_buffer("posterior_mean_coef1", torch.from_numpy(pc).float())

    def round_sigma(self, sigma, return_index=False):
        sigma = torch.as_tensor(sigma)
        s32 = sigma.to(self.u.device).to(torch.float32).reshape(-1, 1)
        index = (s32 - self.u.reshape(1, -1)).abs().argmin(1)  # nearest table entry in float32
        result = index if return_index else self.u[index].to(sigma.dtype)
        return result.reshape(sigma.shape).to(sigma.device)

    def forward(self, x, sigma, class_labels=None, force_fp32=False, **model_kwargs):
        x = x.to(torch.float32)
        sigma = torch.as_tensor(sigma, device=x.device).to(torch.double).reshape(-1, 1, 1, 1)
        c_in = 1 / (sigma ** 2 + 1).sqrt()
        idx = self.round_sigma(sigma.reshape(-1), return_index=True)
        c_noise = (self.M - idx.to(torch.float32)).to(torch.long)
        out = self.model(c_in.to(torch.float32) * x, c_noise.flatten().repeat(x.shape[0]))
        F_x, vars_ = out[:, : self.img_channels], out[:, self.img_channels:]
        pv = self.posterior_variance[c_noise].reshape(-1, 1, 1, 1)
        pc = self.posterior_mean_coef1[c_noise].reshape(-1, 1, 1, 1)
        x0_var = ((vars_ - pv) / pc.pow(2)).clip(min=1e-6)
        D_x = torch.clamp(x + (-sigma) * F_x.to(torch.float32), -1, 1)
        return D_x, x0_var
