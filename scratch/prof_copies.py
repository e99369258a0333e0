import os, sys, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import build_net
dev = torch.device('cuda:0')
net, cfg = build_net("ffhq", dev, "hip")
x = torch.randn(4, 3, 256, 256, device=dev, dtype=torch.float64); sig = torch.tensor(5.0, dtype=torch.float64, device=dev)
def call():
    xt = x.clone().requires_grad_(); D, _ = net(xt, sig); g, = torch.autograd.grad((D * D.detach()).sum(), xt); return g
for _ in range(2): call()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    call(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::add", "aten::mul", "aten::cat", "aten::fill_", "aten::zero_", "aten::zeros", "aten::sub", "aten::div", "aten::silu", "aten::linear", "aten::addmm"):
        st = [f for f in (e.stack or []) if "free-hunch_amd" in f or "free_hunch_amd" in f]
        where = st[0].split("/")[-1] if st else "(outside package)"
        cnt[(e.name, where[:90])] += 1
for (n, w), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print("%4d  %-18s %s" % (c, n, w))
