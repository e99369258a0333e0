import os, sys, time, torch, numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT','.')
sys.path.insert(0, ROOT)
import bench
from concurrent.futures import ThreadPoolExecutor
from free_hunch_amd.measurements import get_operator
from free_hunch_amd.sampler import StandardRGBEncoder, conditional_sampler_batched
dev=torch.device('cuda:0')
net,cfg=bench.build_net('ffhq', dev, 'hip')
images=bench.smooth_images(8,256,1234)
data_dir=os.path.join(ROOT,'free-hunch_amd','data')
enc=StandardRGBEncoder()
def run(groups, seeds):
    ops,ys,noises=[],[],[]
    for b,(img,seed) in enumerate(zip(images,seeds)):
        np.random.seed(seed); torch.manual_seed(seed)
        op=get_operator(name='gaussian_blur',device=dev,sigma_s=0.1,kernel_size=61,intensity=1.0,scale_factor=4,in_shape=(1,3,256,256),mask_opt={"mask_type":"random","mask_len_range":(64,156),"mask_prob_range":(0.6,0.8),"image_size":256})
        op.ctx_slot=b; ops.append(op)
        ys.append(op.forward(enc.encode(img[None].to(dev)),noiseless=False))
        noises.append(torch.randn((1,3,256,256),generator=torch.Generator().manual_seed(seed),dtype=torch.float32))
    B=8; bounds=[round(g*B/groups) for g in range(groups+1)]
    main=torch.cuda.current_stream(); ready=torch.cuda.Event(); ready.record(main)
    def run_group(g):
        lo,hi=bounds[g],bounds[g+1]
        torch.cuda.set_device(0)
        stream=torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            x=conditional_sampler_batched(net,torch.cat(noises[lo:hi],0).to(dev),ys[lo:hi],ops[lo:hi],num_steps=30,sigma_min=0.002,sigma_max=80,rho=7,solver='heun',slot_base=lo,exclusive_device=(groups==1),**bench.fh_kwargs(data_dir,'heun'))
            out=enc.decode(x); out.record_stream(main); done=torch.cuda.Event(); done.record(stream)
        return out,done
    if groups==1: res=[run_group(0)]
    else:
        with ThreadPoolExecutor(max_workers=groups) as pool: res=list(pool.map(run_group,range(groups)))
    for o,d in res: main.wait_event(d)
    return torch.cat([r[0] for r in res],0)
for groups in (2,4,1,2,4,1):
    for it in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter()
        out=run(groups,[it*8+j for j in range(8)])
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print('groups',groups,'iter',it,'%.3f s  %.3f img/s'%(dt,8/dt),flush=True)
