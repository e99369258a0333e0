import sys, os, numpy as np, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden'); sys.path.insert(0,'/root/repo/tests')
from oracle import fh_oracle as fo, unet_oracle as uo
import inputs
from test_oracle_golden import _mk_op
torch.set_num_threads(8)
size=int(sys.argv[2]); name=sys.argv[1]; nsteps=int(sys.argv[3]); solver=sys.argv[4]
kind=sys.argv[5] if len(sys.argv)>5 else 'gauss'
g=np.load('/root/repo/tests/golden/trajectories256.npz' if size==256 else '/root/repo/tests/golden/trajectories.npz')
DATA='/root/repo/free-hunch_amd/data'
import tempfile
if size!=256:
    DATA=tempfile.mkdtemp(); torch.save(torch.from_numpy(g['dct_variance64']), DATA+'/dct_variance.pt')
tag={('inpainting',256):'ip256_heun30__',('gaussian_blur',256):'gb256_heun30__',('inpainting',64):'ip_euler20__',('gaussian_blur',64):'gb_heun10__'}[(name,size)]
y0=torch.from_numpy(g[tag+'y'])
real=fo.LinearPrecond(None) if False else None
u=fo.linear_sigma_table()
class StandIn:
    sigma_min, sigma_max = float(u[-1]) if u[-1]<u[0] else float(u[0]), float(u.max())
    def __init__(s): s.u=u
    def round_sigma(s,x): return fo.round_sigma(u,x)
    def __call__(s,x,sigma):
        return x*(0.25/(0.25+sigma**2)), None
class ScaledUNet:
    def __init__(s, scale):
        cfg=inputs.SMALL_A if size==64 else inputs.SMALL_C
        sd=uo.seeded_state(cfg, 11)
        ok=[k for k in sd if k.startswith('out.')]; print(ok)
        for k in ok:
            if k.startswith('out.2'): sd[k]=sd[k]*scale
        s.net=fo.LinearPrecond(uo.OracleUNet(cfg, sd)); s.pert=0
        s.u=s.net.u; s.sigma_min=s.net.sigma_min; s.sigma_max=s.net.sigma_max; s.scale=scale
    def round_sigma(s,x): return s.net.round_sigma(x)
    def __call__(s,x,sigma):
        d,v=s.net(x,sigma)
        if s.pert: d=d*(1+s.pert*torch.randn(d.shape,generator=torch.Generator().manual_seed(5),dtype=d.dtype))
        return d,v
net=StandIn() if kind=='gauss' else ScaledUNet(float(os.environ.get('DAMP','0.05')))
print('sig range', net.sigma_min, net.sigma_max)
outs=[]
for pert in (0,1e-16):
    op=_mk_op(name,size,g,tag)
    x=inputs.smooth_image(size,int(g[tag+'seeds'][0]))
    if name!='inpainting': op.forward(x.clone())
    noise=inputs.randn((1,3,size,size),int(g[tag+'seeds'][1]),torch.float32)
    y=y0.clone()
    if pert: y=(y.double()*(1+pert*torch.randn(y.shape,generator=torch.Generator().manual_seed(1),dtype=torch.float64)))
    t0=time.time()
    if kind!='gauss': net.pert=1e-6 if pert else 0
    xf,mech=fo.conditional_sampler(net,noise,y,op,num_steps=nsteps,solver=solver,mechanism_factory=lambda op_,v0,d: fo.OracleFreeHunch(1.0,op_,False,v0,d,image_base_covariance='dct_diagonal',data_dir=DATA,max_rtol=float(os.environ.get('MAXRTOL','1'))))
    print('std',[round(float(t.get('std',-1)),3) for t in mech.trace]);print('secs',time.time()-t0,'niter',[t['niter'] for t in mech.trace],'k',mech.trace[-1]['k'],'branches',''.join('c' if t['branch']=='cov' else 'v' for t in mech.trace))
    outs.append(xf)
print('final absmax',float(outs[0].abs().max()),'diff',float((outs[0]-outs[1]).abs().max()))
