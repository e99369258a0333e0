import os, sys, torch, time, collections
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import build_net
from free_hunch_amd import unet_hip
dev=torch.device('cuda:0')
arch=sys.argv[1] if len(sys.argv)>1 else "ffhq"; bs=int(sys.argv[2]) if len(sys.argv)>2 else 8
net,cfg=build_net(arch, dev, "hip")
acc=collections.defaultdict(lambda:[0.0,0,0.0])
def wrap(name, keyfn, flopfn=None):
    orig=getattr(unet_hip.HipOps, name)
    def f(self,*a,**k):
        torch.cuda.synchronize(); t0=time.perf_counter()
        out=orig(self,*a,**k)
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
        key=(name,)+keyfn(self,*a,**k)
        acc[key][0]+=dt; acc[key][1]+=1
        if flopfn: acc[key][2]+=flopfn(self,*a,**k)
        return out
    setattr(unet_hip.HipOps, name, f)
wrap("_conv", lambda s,n,x,**k:(tuple(x.shape), s.conv[n].co, s.conv[n].kh), lambda s,n,x,**k: 2.0*x.shape[0]*x.shape[1]*x.shape[2]*x.shape[3]*s.conv[n].co*s.conv[n].kh*s.conv[n].kw)
wrap("_dgrad", lambda s,n,g,**k:(tuple(g.shape), s.conv[n].ci, s.conv[n].kh), lambda s,n,g,**k: 2.0*g.shape[0]*g.shape[1]*g.shape[2]*s.conv[n].co_p*s.conv[n].ci*s.conv[n].kh*s.conv[n].kw)
wrap("_gn_stats", lambda s,x:(tuple(x.shape),))
wrap("_gn_apply", lambda s,n,x,*a,**k:(tuple(x.shape),))
wrap("_gn_conv", lambda s,g,c,x,*a,**k:(tuple(x.shape), s.conv[c].co), lambda s,g,c,x,*a,**k: 2.0*x.shape[0]*x.shape[1]*x.shape[2]*x.shape[3]*s.conv[c].co*9)
wrap("_attn_fwd", lambda s,p,x,h,t:(tuple(x.shape),))
wrap("_attn_bwd", lambda s,rec,g:(tuple(g.shape),))
wrap("_gn_bwd", lambda s,n,x,st,dy,act,*a,**k:(tuple(x.shape),))
wrap("_resample", lambda s,x,m:(tuple(x.shape),m))
wrap("_add", lambda s,a,b:(tuple(a.shape),))
x = torch.randn(bs,3,256,256, device=dev, dtype=torch.float64); sig=torch.tensor(5.0,dtype=torch.float64,device=dev)
for it in range(2):
    acc.clear()
    xt=x.clone().requires_grad_(); D,_=net(xt,sig); g,=torch.autograd.grad((D*D.detach()).sum(), xt)
tot=sum(v[0] for v in acc.values())
print("sum of op times %.1f ms"%(tot*1e3))
for k,v in sorted(acc.items(), key=lambda kv:-kv[1][0])[:60]:
    print("%-60s n=%3d %7.2f ms  %s" % (str(k), v[1], v[0]*1e3, ("%.0f TF/s"%(v[2]/v[0]/1e12)) if v[2] else ""))
