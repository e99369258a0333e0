"""CPU oracle for the OpenAI guided-diffusion UNet (plain torch fp32 ops, autograd for the VJP).

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  Restates
``training/openai_unet.py:395-686`` (UNetModel), ``:143-256`` (ResBlock),
``:259-305`` (AttentionBlock), ``:328-388`` (both attention orders),
``training/openai_nn.py:17-19,103-121`` (GroupNorm32, timestep_embedding) and
``training/openai_util.py:130-186`` (create_model) as a function of a flat
state dict whose keys are the reference's ``state_dict()`` keys, so that the
same checkpoint drives the reference, this oracle and the HIP product.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    image_size: int = 256
    num_channels: int = 256
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = ()
    learn_sigma: bool = True
    attention_resolutions: str = "32,16,8"
    num_heads: int = 4
    num_head_channels: int = 64
    use_scale_shift_norm: bool = True
    resblock_updown: bool = True
    use_new_attention_order: bool = False
    conv_resample: bool = True
    in_channels: int = 3

    def resolved_mult(self):
        if self.channel_mult:
            return tuple(self.channel_mult)
        return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4),
                64: (1, 2, 3, 4)}[self.image_size]

    def attention_ds(self):
        return tuple(self.image_size // int(r) for r in self.attention_resolutions.split(","))

    @property
    def out_channels(self):
        return 6 if self.learn_sigma else 3


IMAGENET256 = UNetConfig(256, 256, 2, (), True, "32,16,8", 4, 64, True, True, False)
FFHQ256 = UNetConfig(256, 128, 1, (), True, "16", 4, 64, True, True, False)


def layout(cfg: UNetConfig):
    """Walk the constructor (openai_unet.py:479-611) and return the block structure:
    list of (prefix, kind, cin, cout, heads) with kind in {conv_in, res, res_down, res_up, attn, down, up}."""
    mc, mult, ads = cfg.num_channels, cfg.resolved_mult(), cfg.attention_ds()
    heads = lambda ch: (cfg.num_heads if cfg.num_head_channels == -1 else ch // cfg.num_head_channels)
    ch = int(mult[0] * mc)
    inp = [[("input_blocks.0.0", "conv_in", cfg.in_channels, ch, 0)]]
    chans, ds = [ch], 1
    for level, m in enumerate(mult):
        for _ in range(cfg.num_res_blocks):
            blk = [(f"input_blocks.{len(inp)}.0", "res", ch, int(m * mc), 0)]
            ch = int(m * mc)
            if ds in ads:
                blk.append((f"input_blocks.{len(inp)}.1", "attn", ch, ch, heads(ch)))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            kind = "res_down" if cfg.resblock_updown else "down"
            inp.append([(f"input_blocks.{len(inp)}.0", kind, ch, ch, 0)])
            chans.append(ch)
            ds *= 2
    mid = [("middle_block.0", "res", ch, ch, 0), ("middle_block.1", "attn", ch, ch, heads(ch)),
           ("middle_block.2", "res", ch, ch, 0)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            p = f"output_blocks.{len(out)}"
            blk = [(f"{p}.0", "res", ch + ich, int(mc * m), 0)]
            ch = int(mc * m)
            if ds in ads:
                blk.append((f"{p}.{len(blk)}", "attn", ch, ch, heads(ch)))
            if level and i == cfg.num_res_blocks:
                kind = "res_up" if cfg.resblock_updown else "up"
                blk.append((f"{p}.{len(blk)}", kind, ch, ch, 0))
                ds //= 2
            out.append(blk)
    return inp, mid, out, int(mult[0] * mc)


def state_shapes(cfg: UNetConfig):
    """{key: shape} of the reference state_dict for this config."""
    sh = {}
    mc = cfg.num_channels
    ted = mc * 4
    sh["time_embed.0.weight"], sh["time_embed.0.bias"] = (ted, mc), (ted,)
    sh["time_embed.2.weight"], sh["time_embed.2.bias"] = (ted, ted), (ted,)

    def conv(p, co, ci, k):
        sh[p + ".weight"], sh[p + ".bias"] = (co, ci, k, k), (co,)

    def gn(p, c):
        sh[p + ".weight"], sh[p + ".bias"] = (c,), (c,)

    inp, mid, out, ch0 = layout(cfg)
    for blk in inp + [mid] + out:
        for (p, kind, ci, co, _h) in blk:
            if kind == "conv_in":
                conv(p, co, ci, 3)
            elif kind in ("res", "res_down", "res_up"):
                gn(p + ".in_layers.0", ci)
                conv(p + ".in_layers.2", co, ci, 3)
                eo = 2 * co if cfg.use_scale_shift_norm else co
                sh[p + ".emb_layers.1.weight"], sh[p + ".emb_layers.1.bias"] = (eo, ted), (eo,)
                gn(p + ".out_layers.0", co)
                conv(p + ".out_layers.3", co, co, 3)
                if ci != co:
                    conv(p + ".skip_connection", co, ci, 1)
            elif kind == "attn":
                gn(p + ".norm", ci)
                sh[p + ".qkv.weight"], sh[p + ".qkv.bias"] = (3 * ci, ci, 1), (3 * ci,)
                sh[p + ".proj_out.weight"], sh[p + ".proj_out.bias"] = (ci, ci, 1), (ci,)
            elif kind == "down" and cfg.conv_resample:
                conv(p + ".op", co, ci, 3)
            elif kind == "up" and cfg.conv_resample:
                conv(p + ".conv", co, ci, 3)
    gn("out.0", ch0)
    conv("out.2", cfg.out_channels, ch0, 3)
    return sh


def seeded_state(cfg: UNetConfig, seed: int, scale: float = 1.0):
    """Deterministic synthetic weights (no checkpoint exists offline).  Keys visited in sorted order, one
    CPU generator; conv/linear ~ N(0, 1/fan_in)*scale, GN weight ~ 1 + 0.1 N, biases ~ 0.02 N.  Layers the
    reference zero-initialises (openai_nn.py:68-74) are random here as well, otherwise the UNet outputs 0."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in sorted(state_shapes(cfg).items()):
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if k.endswith(".bias"):
            sd[k] = 0.02 * r
        elif len(shp) == 1:
            sd[k] = 1.0 + 0.1 * r
        else:
            fan_in = int(torch.tensor(shp[1:]).prod())
            sd[k] = r * (scale / math.sqrt(fan_in))
    return sd


def timestep_embedding(t, dim, max_period=10000):  # openai_nn.py:103-121
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, p, x):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _res(sd, p, kind, x, emb, cfg):
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if kind == "res_down":
        h, x = F.avg_pool2d(h, 2), F.avg_pool2d(x, 2)
    elif kind == "res_up":
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[..., None, None]
    if cfg.use_scale_shift_norm:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = F.silu(_gn(sd, p + ".out_layers.0", h + e))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def _attn(sd, p, x, n_heads, new_order):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    T = xf.shape[-1]
    ch = c // n_heads
    if new_order:  # QKVAttention :361-388
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (t.reshape(b * n_heads, ch, T) for t in (q, k, v))
    else:          # QKVAttentionLegacy :328-354
        q, k, v = qkv.reshape(b * n_heads, ch * 3, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, T)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def _run(sd, blk, h, emb, cfg):
    for (p, kind, _ci, _co, heads) in blk:
        if kind == "conv_in":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind in ("res", "res_down", "res_up"):
            h = _res(sd, p, kind, h, emb, cfg)
        elif kind == "attn":
            h = _attn(sd, p, h, heads, cfg.use_new_attention_order)
        elif kind == "down":
            h = (F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
                 if cfg.conv_resample else F.avg_pool2d(h, 2))
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if cfg.conv_resample:
                h = F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
    return h


def unet_forward(sd, cfg: UNetConfig, x, timesteps):
    """UNetModel.forward (openai_unet.py:648-686), unconditional."""
    emb = timestep_embedding(timesteps, cfg.num_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    inp, mid, out, _ = layout(cfg)
    hs, h = [], x.float()
    for blk in inp:
        h = _run(sd, blk, h, emb, cfg)
        hs.append(h)
    h = _run(sd, mid, h, emb, cfg)
    for blk in out:
        h = _run(sd, blk, torch.cat([h, hs.pop()], dim=1), emb, cfg)
    h = F.silu(_gn(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


class OracleUNet:
    def __init__(self, cfg: UNetConfig, sd):
        self.cfg, self.sd = cfg, sd

    def __call__(self, x, timesteps):
        return unet_forward(self.sd, self.cfg, x, timesteps)
