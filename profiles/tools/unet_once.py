"""Three UNet forward + input-VJP calls at batch 8 (for `rocprofv3 --kernel-trace`; see unet_trace.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import build_net
dev = torch.device('cuda:0')
arch = sys.argv[1] if len(sys.argv) > 1 else "ffhq"; bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
net, cfg = build_net(arch, dev, "hip", os.environ.get("FH_UNET_DTYPE", "fp32"))
x = torch.randn(bs, 3, 256, 256, device=dev, dtype=torch.float64); sig = torch.tensor(5.0, dtype=torch.float64, device=dev)
for it in range(3):
    xt = x.clone().requires_grad_(); D, _ = net(xt, sig); g, = torch.autograd.grad((D * D.detach()).sum(), xt)
torch.cuda.synchronize()
