import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from free_hunch_amd import _lib
dev=torch.device('cuda:0'); S=256; d=3*S*S
ctx=_lib.Context.get(S,3,128)
g=torch.Generator().manual_seed(0)
B=torch.randn(64,d,generator=g,dtype=torch.float64).to(dev)
Dx=(torch.rand(d,generator=g,dtype=torch.float64)+0.5).to(dev); rx=(torch.rand(d,generator=g,dtype=torch.float64)+0.5).to(dev)
Dy=torch.empty_like(Dx); ry=torch.empty_like(rx); G=torch.zeros(128,128,dtype=torch.float64,device=dev)
for m in (2,8,16,32,48,64):
    for _ in range(3): ctx.rep_invert(Dx,rx,B,0.0,Dy,ry,G,m)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ctx.rep_invert(Dx,rx,B,0.0,Dy,ry,G,m)
    e1.record(); torch.cuda.synchronize()
    w=rx*rx/Dx
    ref=(B[:m]*w)@B[:m].T
    err=float((G[:m,:m]-ref).abs().max()/ref.abs().max())
    print("m=%d  fh_rep_invert %.1f us  rel err %.1e"%(m, e0.elapsed_time(e1)*1e3/20, err), flush=True)
