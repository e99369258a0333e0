import sys, time, torch, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests/golden']
from oracle import fh_oracle as fo
import inputs
torch.set_num_threads(8)
DATA='/root/repo/free-hunch_amd/data'
shape=(1,3,256,256)
cov=fo.make_covariance("dct_diagonal", DATA, 80.0**2, 3*256*256)
steps=inputs.script(1, shape, 28, 10.0, sig_end=1.0)
t_all=time.time()
for i,(what,a) in enumerate(steps):
    t0=time.time()
    if what=='time': cov.update_time_step(a['x'],a['sigma'],a['sigma_next'],a['score'])
    else: cov.update_space_step(a['m0'],a['m1'],a['sigma'],a['x'],a['xn'])
    if i%8<2: print(i,what,cov.k,round(time.time()-t0,2),flush=True)
print('total',time.time()-t_all)
