import sys, torch, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests/golden','/root/repo/tests']
import inputs, nets
from oracle import fh_oracle as fo
g=np.load('/root/repo/tests/golden/baselines_tmpd_pos.npz')
for damp in (0.05, 0.01, 0.002):
    inputs.DAMP=damp
    from oracle import unet_oracle as uo
    net=fo.LinearPrecond(uo.OracleUNet(inputs.SMALL_A, inputs.damped_state(uo.seeded_state, inputs.SMALL_A, 11, damp)))
    for sig in (80.0, 20.0, 5.0, 1.0, 0.2):
        x=(inputs.randn((1,3,64,64), 751, torch.float32).double()*sig).requires_grad_()
        D,_=net(x, torch.tensor(sig,dtype=torch.float64))
        (gr,)=torch.autograd.grad(D.sum(), x)
        v=gr*sig**2
        print(damp, sig, 'var min %.3g max %.3g frac_neg %.3g frac_zero %.3g'%(float(v.min()),float(v.max()),float((v<0).float().mean()),float((v==0).float().mean())))
