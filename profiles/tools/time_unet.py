import os, sys, torch, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import build_net
dev=torch.device('cuda:0')
arch=sys.argv[1] if len(sys.argv)>1 else "ffhq"; bs=int(sys.argv[2]) if len(sys.argv)>2 else 8
net,cfg=build_net(arch, dev, "hip")
x = torch.randn(bs,3,256,256, device=dev, dtype=torch.float64); sig=torch.tensor(5.0,dtype=torch.float64,device=dev)
for fuse in ("1","0","1","0"):
    os.environ["FH_GN_FUSE"]=fuse
    tf=tb=0.0
    for it in range(6):
        torch.cuda.synchronize(); t0=time.perf_counter()
        xt=x.clone().requires_grad_(); D,_=net(xt,sig)
        torch.cuda.synchronize(); t1=time.perf_counter()
        g,=torch.autograd.grad((D*D.detach()).sum(), xt)
        torch.cuda.synchronize(); t2=time.perf_counter()
        if it>=2: tf+=t1-t0; tb+=t2-t1
    print("FH_GN_FUSE=%s  fwd %.2f ms  vjp %.2f ms"%(fuse, tf/4*1e3, tb/4*1e3), flush=True)
