import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden'); sys.path.insert(0,'/root/repo/tests')
from oracle import fh_oracle as fo
import inputs
torch.set_num_threads(8)
g=np.load('/root/repo/tests/golden/trajectories256.npz')
DATA='/root/repo/free-hunch_amd/data'
size=256
name=sys.argv[1]
p={'inpainting':'ip256_heun30__','gaussian_blur':'gb256_heun30__'}[name]
y=torch.from_numpy(g[p+'y'])
from test_oracle_golden import _mk_op
op=_mk_op(name,size,g,p)
x=inputs.smooth_image(size,int(g[p+'seeds'][0]))
if name!='inpainting': op.forward(x.clone())
cov=fo.make_covariance("dct_diagonal", DATA, 80.0**2, 3*size*size)
x0=(0.2*x).double()
# CG with iterate recording
def run(perturb, c128=False, maxiter=260):
    C=cov.denoiser_cov_vector_dot
    s2=op.sigma_s.clip(min=0.001)**2
    if name=='inpainting':
        mask=op.mask
        A=lambda u: (s2*u.reshape(x0.shape)+mask*C(mask*u.reshape(x0.shape))).flatten()
        b=(mask*y-mask*x0).flatten()
    else:
        FB,FBC,_,_=op.pre_calculated
        if c128:
            FB=fo.p2o(op._k().double(),(size,size)); FBC=FB.conj()
        blur=lambda v: torch.fft.ifft2(FB*torch.fft.fft2(v)).real
        blur_t=lambda v: torch.fft.ifft2(FBC*torch.fft.fft2(v)).real
        A=lambda u:(s2*u.reshape(y.shape)+blur(C(blur_t(u.reshape(y.shape))))).flatten()
        b=(y-blur(x0)).flatten()
    if perturb:
        gen=torch.Generator().manual_seed(3)
        b=b*(1+perturb*torch.randn(b.shape,generator=gen,dtype=b.dtype))
    xs=[];rn=[]
    xk=b; r=b-A(xk); pk=r.clone(); rz=torch.dot(r,r)
    for k in range(maxiter):
        Ap=A(pk); pAp=torch.dot(pk,Ap); al=rz/pAp
        xk=xk+al*pk; r=r-al*Ap; rzn=torch.dot(r,r); pk=r+(rzn/rz)*pk; rz=rzn
        xs.append(xk.clone()); rn.append(float(rzn.sqrt()/b.norm()))
    return xs,rn
a,ra=run(0)
b_,rb=run(1e-16)
outs=[('ulp',b_,rb)]
if name!='inpainting':
    c,rc=run(0,True); outs.append(('c128',c,rc))
for lab,o,ro in outs:
    print(lab)
    for k in [0,1,2,4,9,19,39,59,79,99,119,149,199,259]:
        d=float((a[k]-o[k]).abs().max()/a[k].abs().max())
        print(k+1,'rel diff %.3g  resid %.4g %.4g'%(d,ra[k],ro[k]))
