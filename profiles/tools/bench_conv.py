import os, sys, torch, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from free_hunch_amd import _lib as L
lib=L.load(); dev=torch.device('cuda:0')
shapes=[(1,256,256,128,128,3),(8,256,256,128,128,3),(1,256,256,256,128,3),(1,128,128,128,128,3),(1,64,64,256,256,3),(8,64,64,256,256,3),
        (1,32,32,256,256,3),(1,16,16,512,512,3),(8,16,16,512,512,3),(1,8,8,512,512,3),(8,8,8,1024,512,3),(1,16,16,512,1536,1),(1,256,256,256,256,3),(1,64,64,512,512,3)]
for (N,H,W,Ci,Co,k) in shapes:
    x=torch.randn(N,H,W,Ci,device=dev); w=torch.randn(Co,k*k,Ci,device=dev)*0.05; b=torch.zeros(Co,device=dev); out=torch.empty(N,H,W,Co,device=dev)
    ks=lib.fh_conv2d_splitk(N,H,W,Ci,Co,k,k); ws=torch.empty(max(ks,1),N*H*W,Co,device=dev)
    f=lambda: L.check(lib.fh_conv2d_nhwc(x.data_ptr(),w.data_ptr(),b.data_ptr(),None,out.data_ptr(),ws.data_ptr(),ks,N,H,W,Ci,Co,k,k,k//2,1,L.stream()),"c")
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    it=10; e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/it
    fl=2.0*N*H*W*Ci*Co*k*k
    print("N%d %3dx%-3d Ci%4d Co%4d k%d : %8.3f ms  %6.1f TF/s" % (N,H,W,Ci,Co,k,ms,fl/ms/1e9), flush=True)
