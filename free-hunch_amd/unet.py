"""OpenAI guided-diffusion UNet for the Free Hunch path (reference: training/openai_unet.py:395-686,
openai_nn.py:17-121, openai_util.py:130-186, openai_loading_utils.py:12-42).

`UNetModel` holds the reference's state-dict keys verbatim, so `load_state_dict` accepts the public
256x256 ImageNet / FFHQ checkpoints unchanged.  The network is compiled once into a flat list of steps
(conv3x3, GroupNorm+SiLU(+scale/shift), attention, up/down-sample, skip push/pop); each step runs on the
backend selected at construction:

  backend="hip"    hand-written gfx950 kernels (libfh_hip.so) wrapped as autograd Functions (forward and
                   input-gradient); activations stay NHWC float32 in HBM.
  backend="torch"  PyTorch-ROCm ops - the bring-up / A-B path used by the per-block parity tests only.
"""
from __future__ import annotations

import argparse
import math
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


_PRECISION_MODE = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16": 3, "fp16x3": 4}  # fh_unet_set_precision codes


@dataclass
class UNetConfig:
    image_size: int = 256
    num_channels: int = 256
    num_res_blocks: int = 2
    channel_mult: Tuple[float, ...] = ()
    learn_sigma: bool = True
    attention_resolutions: str = "32,16,8"
    num_heads: int = 4
    num_head_channels: int = 64
    use_scale_shift_norm: bool = True
    resblock_updown: bool = True
    use_new_attention_order: bool = False
    conv_resample: bool = True
    in_channels: int = 3

    def mult(self):
        if self.channel_mult:
            return tuple(self.channel_mult)
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if self.image_size not in table:
            raise ValueError(f"unsupported image size: {self.image_size}")
        return table[self.image_size]

    @property
    def out_channels(self):
        return 6 if self.learn_sigma else 3


# architectures of the two public checkpoints the reference's README names
IMAGENET256 = UNetConfig(256, 256, 2, (), True, "32,16,8", 4, 64, True, True, False)
FFHQ256 = UNetConfig(256, 128, 1, (), True, "16", 4, 64, True, True, False)


def config_from_setup_text(text):
    """Parse the reference's `models/*_setup.txt` flag string (openai_loading_utils.py:5-39)."""
    kv = {}
    for part in text.strip().split("--")[1:]:
        key, value = part.strip().split(" ", 1)
        kv[key] = value.strip()
    tb = lambda k, d="False": kv.get(k, d).lower() == "true"
    if tb("class_cond"):
        raise NotImplementedError("class-conditional checkpoints are outside the Free Hunch CLI path")
    cm = kv.get("channel_mult", "")
    return UNetConfig(image_size=int(kv["image_size"]), num_channels=int(kv["num_channels"]),
                      num_res_blocks=int(kv["num_res_blocks"]),
                      channel_mult=tuple(int(c) for c in cm.split(",")) if cm else (),
                      learn_sigma=tb("learn_sigma"), attention_resolutions=kv.get("attention_resolutions", "16"),
                      num_heads=int(kv.get("num_heads", 1)), num_head_channels=int(kv.get("num_head_channels", -1)),
                      use_scale_shift_norm=tb("use_scale_shift_norm"), resblock_updown=tb("resblock_updown"),
                      use_new_attention_order=tb("use_new_attention_order")), tb("use_fp16")


def _plan(cfg: UNetConfig):
    """Flatten the constructor of the reference (openai_unet.py:479-611) into steps.
    Step = (op, prefix, cin, cout, heads);  ops: conv_in res res_down res_up attn down up push pop_cat."""
    mc, mult = cfg.num_channels, cfg.mult()
    att = {cfg.image_size // int(r) for r in cfg.attention_resolutions.split(",")}
    nheads = lambda c: cfg.num_heads if cfg.num_head_channels == -1 else c // cfg.num_head_channels
    steps, skip = [], []
    ch = int(mult[0] * mc)
    steps += [("conv_in", "input_blocks.0.0", cfg.in_channels, ch, 0), ("push", "", 0, 0, 0)]
    skip.append(ch)
    idx, ds = 1, 1
    for level, m in enumerate(mult):
        for _ in range(cfg.num_res_blocks):
            co = int(m * mc)
            steps.append(("res", f"input_blocks.{idx}.0", ch, co, 0))
            ch = co
            if ds in att:
                steps.append(("attn", f"input_blocks.{idx}.1", ch, ch, nheads(ch)))
            steps.append(("push", "", 0, 0, 0))
            skip.append(ch)
            idx += 1
        if level != len(mult) - 1:
            steps.append(("res_down" if cfg.resblock_updown else "down", f"input_blocks.{idx}.0", ch, ch, 0))
            steps.append(("push", "", 0, 0, 0))
            skip.append(ch)
            idx += 1
            ds *= 2
    steps += [("res", "middle_block.0", ch, ch, 0), ("attn", "middle_block.1", ch, ch, nheads(ch)),
              ("res", "middle_block.2", ch, ch, 0)]
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = skip.pop()
            co = int(mc * m)
            steps.append(("pop_cat", "", 0, 0, 0))
            steps.append(("res", f"output_blocks.{idx}.0", ch + ich, co, 0))
            ch = co
            sub = 1
            if ds in att:
                steps.append(("attn", f"output_blocks.{idx}.{sub}", ch, ch, nheads(ch)))
                sub += 1
            if level and i == cfg.num_res_blocks:
                steps.append(("res_up" if cfg.resblock_updown else "up", f"output_blocks.{idx}.{sub}", ch, ch, 0))
                ds //= 2
            idx += 1
    return steps, int(mult[0] * mc)


def parameter_shapes(cfg: UNetConfig):
    mc, ted = cfg.num_channels, cfg.num_channels * 4
    sh = {"time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
          "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}

    def conv(p, co, ci, k):
        sh[p + ".weight"], sh[p + ".bias"] = (co, ci, k, k), (co,)

    def vec(p, c):
        sh[p + ".weight"], sh[p + ".bias"] = (c,), (c,)

    steps, ch0 = _plan(cfg)
    for op, p, ci, co, _ in steps:
        if op == "conv_in":
            conv(p, co, ci, 3)
        elif op.startswith("res"):
            vec(p + ".in_layers.0", ci)
            conv(p + ".in_layers.2", co, ci, 3)
            eo = 2 * co if cfg.use_scale_shift_norm else co
            sh[p + ".emb_layers.1.weight"], sh[p + ".emb_layers.1.bias"] = (eo, ted), (eo,)
            vec(p + ".out_layers.0", co)
            conv(p + ".out_layers.3", co, co, 3)
            if ci != co:
                conv(p + ".skip_connection", co, ci, 1)
        elif op == "attn":
            vec(p + ".norm", ci)
            sh[p + ".qkv.weight"], sh[p + ".qkv.bias"] = (3 * ci, ci, 1), (3 * ci,)
            sh[p + ".proj_out.weight"], sh[p + ".proj_out.bias"] = (ci, ci, 1), (ci,)
        elif op == "down" and cfg.conv_resample:
            conv(p + ".op", co, ci, 3)
        elif op == "up" and cfg.conv_resample:
            conv(p + ".conv", co, ci, 3)
    vec("out.0", ch0)
    conv("out.2", cfg.out_channels, ch0, 3)
    return sh


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class _TorchOps:
    """PyTorch-ROCm execution of the step list (bring-up / A-B path; NCHW)."""

    def __init__(self, cfg, P):
        self.cfg, self.P = cfg, P

    def gn(self, p, x):
        return F.group_norm(x.float(), 32, self.P[p + ".weight"], self.P[p + ".bias"], eps=1e-5)

    def conv(self, p, x, stride=1):
        w = self.P[p + ".weight"]
        return F.conv2d(x, w, self.P[p + ".bias"], stride=stride, padding=w.shape[-1] // 2)

    def res(self, op, p, x, emb):
        P, cfg = self.P, self.cfg
        h = F.silu(self.gn(p + ".in_layers.0", x))
        if op == "res_down":
            h, x = F.avg_pool2d(h, 2), F.avg_pool2d(x, 2)
        elif op == "res_up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        h = self.conv(p + ".in_layers.2", h)
        e = F.linear(F.silu(emb), P[p + ".emb_layers.1.weight"], P[p + ".emb_layers.1.bias"])[..., None, None]
        if cfg.use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = F.silu(self.gn(p + ".out_layers.0", h) * (1 + scale) + shift)
        else:
            h = F.silu(self.gn(p + ".out_layers.0", h + e))
        h = self.conv(p + ".out_layers.3", h)
        if p + ".skip_connection.weight" in P:
            x = self.conv(p + ".skip_connection", x)
        return x + h

    def attn(self, p, x, heads):
        P = self.P
        b, c, hh, ww = x.shape
        xf = x.reshape(b, c, -1)
        qkv = F.conv1d(self.gn(p + ".norm", xf), P[p + ".qkv.weight"], P[p + ".qkv.bias"])
        T, ch = xf.shape[-1], c // heads
        if self.cfg.use_new_attention_order:
            q, k, v = (t.reshape(b * heads, ch, T) for t in qkv.chunk(3, dim=1))
        else:
            q, k, v = qkv.reshape(b * heads, ch * 3, T).split(ch, dim=1)
        s = 1 / math.sqrt(math.sqrt(ch))
        w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
        a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, T)
        return (xf + F.conv1d(a, P[p + ".proj_out.weight"], P[p + ".proj_out.bias"])).reshape(b, c, hh, ww)

    def run(self, steps, x, emb, emb_key=None):
        cfg, h, stack = self.cfg, x, []
        for op, p, _ci, _co, heads in steps:
            if op == "push":
                stack.append(h)
            elif op == "pop_cat":
                h = torch.cat([h, stack.pop()], dim=1)
            elif op == "conv_in":
                h = self.conv(p, h)
            elif op in ("res", "res_down", "res_up"):
                h = self.res(op, p, h, emb)
            elif op == "attn":
                h = self.attn(p, h, heads)
            elif op == "down":
                h = self.conv(p + ".op", h, stride=2) if cfg.conv_resample else F.avg_pool2d(h, 2)
            elif op == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                if cfg.conv_resample:
                    h = self.conv(p + ".conv", h)
        h = F.silu(self.gn("out.0", h))
        return self.conv("out.2", h)


class UNetModel(torch.nn.Module):
    """forward(x [N,3,H,W] float32, timesteps [N]) -> [N, out_channels, H, W]; state-dict keys = the reference's."""

    def __init__(self, cfg: UNetConfig, backend="hip", dtype="fp32"):
        super().__init__()
        self.cfg, self.backend = cfg, backend
        self.set_dtype(dtype)
        self.steps, self.ch0 = _plan(cfg)
        self._names = {}
        for k, shp in parameter_shapes(cfg).items():
            safe = k.replace(".", "__")
            self._names[k] = safe
            self.register_parameter(safe, torch.nn.Parameter(torch.zeros(shp), requires_grad=False))
        self.img_channels, self.img_resolution, self.label_dim = cfg.in_channels, cfg.image_size, 0
        self.sigma_min, self.sigma_max = 0.0, 1e20
        self._ops = None
        self._emb = {}

    def set_dtype(self, dtype):
        """"fp32" (default): every convolution has fp32 accuracy (exact 3-way bf16 split on the matrix cores).
        "fp16": the reference's `use_fp16` torso (training/openai_fp16_util.py:15-32, openai_unet.py:464, 625-638): the
        convolutions of the input / middle / output blocks multiply operands rounded to IEEE half precision (weights once,
        activations while they are staged) on v_mfma_f32_32x32x16_f16 with fp32 accumulation; GroupNorm, softmax, the time
        embedding and the first / last layers stay fp32 as in the reference, and so does the STORAGE between layers (the
        reference stores half there: one more rounding per tensor).  Pinned to the reference's own fp16 output
        (tests/golden/unet_a_fp16.npz).  Outside the fp32 parity bar - reported as a separate mode.
        "bf16": the same with operands rounded to bfloat16 (8-bit significand, fp32 exponent range; same MFMA rate).
        "bf16x3": operands carried as two bf16 planes, three matrix products per convolution (relative error ~ 2^-16, i.e.
        between TF32 - what the reference's convolutions run in by default on its CUDA path - and fp32) at half the
        matrix work of "fp32".  Also reported separately.
        "fp16x3" (half-split): the 3 x 3 convolutions carry both operands as TWO half-precision planes of the power-of-two
        scaled value (h + m = x to within one fp32 ulp) and form h h' + h m' + m h' on the f16 matrix cores: half the matrix
        work of "fp32"; the dropped m m' term (2^-24 rms of a product) sits below the rounding noise of the fp32 accumulation
        that any fp32 convolution - the reference's included - carries, so against float64 the error is that of an fp32
        convolution (tests/test_hip_unet.py::test_half_split_mode_*).  Not bit-comparable with the exact split, hence opt-in."""
        if dtype not in _PRECISION_MODE:
            raise ValueError(f"unet dtype must be fp32, fp16x3, bf16x3, bf16 or fp16, got {dtype}")
        if dtype != "fp32" and self.backend != "hip":
            raise NotImplementedError("the reduced-precision torso exists on the hip backend only")
        self.dtype_mode = dtype
        if getattr(self, "_ops", None) is not None and hasattr(self._ops, "bf16"):
            self._ops.bf16 = _PRECISION_MODE[dtype]
        return self

    # the reference's key names in and out ------------------------------------------------------------
    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        back = {v: kk for kk, v in self._names.items()}
        return {back[name]: t for name, t in sd.items()}

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._names if k not in sd]
        extra = [k for k in sd if k not in self._names]
        if strict and (missing or extra):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:4]}..., unexpected {extra[:4]}...")
        with torch.no_grad():
            for k, safe in self._names.items():
                if k in sd:
                    getattr(self, safe).copy_(sd[k])
        self._ops = None
        self._emb = {}
        return self

    def _params(self):
        return {k: getattr(self, safe) for k, safe in self._names.items()}

    def _backend_ops(self):
        if self._ops is None:
            if self.backend == "torch":
                self._ops = _TorchOps(self.cfg, self._params())
            elif self.backend == "hip":
                from .unet_hip import HipOps
                self._ops = HipOps(self.cfg, self._params())
                self._ops.bf16 = _PRECISION_MODE[self.dtype_mode]
            else:
                raise ValueError(f"unknown backend {self.backend}")
        return self._ops

    def _apply(self, fn, *a, **k):  # moving / casting the module invalidates prepared weights
        self._ops = None
        self._emb = {}
        return super()._apply(fn, *a, **k)

    def forward(self, x, timesteps, y=None, class_labels=None):
        assert y is None and class_labels is None, "unconditional checkpoints only"
        P = self._params()
        ops = self._backend_ops()
        key = tuple(timesteps.tolist())  # the embedding MLPs depend on the timestep only: cached per sigma
        emb = self._emb.get(key)
        if emb is None:
            emb = timestep_embedding(timesteps, self.cfg.num_channels)
            emb = F.linear(emb, P["time_embed.0.weight"], P["time_embed.0.bias"])
            emb = F.linear(F.silu(emb), P["time_embed.2.weight"], P["time_embed.2.bias"])
            if len(self._emb) > 4096:
                self._emb.clear()
            self._emb[key] = emb
        return ops.run(self.steps, x.float(), emb, key)


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False,
                 use_fp16=False, use_new_attention_order=False, backend="hip", dtype=None):
    """Same keywords as training/openai_util.py:130 (dropout / checkpointing are inference no-ops)."""
    if class_cond:
        raise NotImplementedError("class-conditional UNets are outside the Free Hunch CLI path")
    cm = tuple(int(c) for c in channel_mult.split(",")) if channel_mult else ()
    cfg = UNetConfig(image_size, num_channels, num_res_blocks, cm, learn_sigma, attention_resolutions, num_heads,
                     num_head_channels, use_scale_shift_norm, resblock_updown, use_new_attention_order)
    return UNetModel(cfg, backend=backend, dtype=dtype if dtype is not None else ("fp16" if use_fp16 else "fp32"))


def load_model(state_dict_path, setup_path, backend="hip", dtype=None):
    """openai_loading_utils.load_model: weights_only state dict + flags text file."""
    sd = torch.load(state_dict_path, map_location="cpu", weights_only=True)
    with open(setup_path) as f:
        cfg, fp16 = config_from_setup_text(f.read())
    # `use_fp16 True` in the setup file selects the reduced-precision torso, as in the reference (openai_loading_utils.py)
    model = UNetModel(cfg, backend=backend, dtype=dtype if dtype is not None else ("fp16" if fp16 else "fp32"))
    model.load_state_dict(sd)
    return model, cfg


def seeded_state(cfg: UNetConfig, seed, scale=1.0):
    """Synthetic weights for benchmarks and tests (no checkpoint is available offline): keys in sorted order from
    one CPU generator; conv/linear ~ N(0, scale^2/fan_in), GroupNorm weight 1 + 0.1 N, biases 0.02 N.  Layers the
    reference zero-initialises are random too, otherwise the UNet outputs zero."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in sorted(parameter_shapes(cfg).items()):
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if k.endswith(".bias"):
            sd[k] = 0.02 * r
        elif len(shp) == 1:
            sd[k] = 1.0 + 0.1 * r
        else:
            sd[k] = r * (scale / math.sqrt(int(torch.tensor(shp[1:]).prod())))
    return sd
