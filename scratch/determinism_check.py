"""Does the lock-step sampler with the HIP UNet depend on what freed device memory contains?  Runs the configuration of
tests/test_timed_path.py::test_batched8_256_equals_per_image_hip_unet several times; between the runs the allocator's free blocks are
poisoned with NaN / huge values.  Prints the CG iteration lists and output checksums of every run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, inputs, nets
from test_hip_parity import _base_kwargs
from test_timed_path import _batch_inputs, DATA
from free_hunch_amd.sampler import conditional_sampler, conditional_sampler_batched
dev = torch.device("cuda:0")
B, S = 8, 256
net = nets.damped_hip_net(inputs.SMALL_C, 13, dev)
kw = _base_kwargs(DATA, {"max_rtol": 1e-6})
ops, ys, noise = _batch_inputs(B, S, dev, "inpainting")
run = dict(num_steps=6, sigma_min=0.002, sigma_max=80, rho=7, solver="heun")

def poison(val):
    junk = [torch.full((n,), val, dtype=torch.float32, device=dev) for n in (1 << 28, 1 << 27, 1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20) for _ in range(3)]
    torch.cuda.synchronize(); del junk

ref = None
for it, val in enumerate([None, float("nan"), 1e30, None, float("nan")]):
    if val is not None: poison(val)
    if it == 3: torch.cuda.empty_cache()
    xb = conditional_sampler_batched(net, noise, ys, ops, **run, **kw)
    n0 = [t["niter"] for t in conditional_sampler_batched.last_mechanisms[0].trace]
    cs = float(xb.double().sum())
    x1, _, _ = conditional_sampler(net, noise[:1], None, None, measurement=ys[0], operator=ops[0], **run, **kw)
    n1 = [t["niter"] for t in conditional_sampler.last_mechanism.trace]
    print(f"run {it} poison={val}: batched image0 niter {n0} checksum {cs:.12e} finite {bool(torch.isfinite(xb).all())} | per-image niter {n1} checksum {float(x1.double().sum()):.12e}", flush=True)
    if ref is None: ref = (xb.clone(), x1.clone())
    else: print(f"      max|batched - run0| = {float((xb-ref[0]).abs().max()):.3e}   max|per-image - run0| = {float((x1-ref[1]).abs().max()):.3e}", flush=True)
