import os, sys, torch, math
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
from free_hunch_amd import _lib as L
from test_hip_unet import wino_weights
lib=L.load(); dev=torch.device('cuda:0')
shapes=[(8,256,256,128,128),(8,256,256,256,128),(8,128,128,256,256),(8,64,64,256,256),(1,256,256,128,128),(8,32,32,256,256),(8,16,16,512,512)]
for (N,H,W,Ci,Co) in shapes:
    x=torch.randn(N,H,W,Ci,device=dev); w=torch.randn(Co,Ci,3,3,device=dev)*0.03; b=torch.zeros(Co,device=dev); out=torch.empty(N,H,W,Co,device=dev)
    wu=wino_weights(w); wd=w.permute(0,2,3,1).reshape(Co,9,Ci).contiguous()
    f1=lambda: L.check(lib.fh_conv3x3_wino_nhwc(x.data_ptr(),wu.data_ptr(),b.data_ptr(),None,out.data_ptr(),N,H,W,Ci,Co,L.stream()),"w")
    ks=lib.fh_conv2d_splitk(N,H,W,Ci,Co,3,3); ws=torch.empty(max(ks,1),N*H*W,Co,device=dev)
    f2=lambda: L.check(lib.fh_conv2d_nhwc(x.data_ptr(),wd.data_ptr(),b.data_ptr(),None,out.data_ptr(),ws.data_ptr(),ks,N,H,W,Ci,Co,3,3,1,1,L.stream()),"c")
    res=[]
    for f in (f1,f2):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/10)
    fl=2.0*N*H*W*Ci*Co*9
    print("N%d %3dx%-3d Ci%4d Co%4d : wino %7.3f ms (%5.1f direct-equiv TF/s)   direct %7.3f ms (%5.1f TF/s)  speedup %.2f" % (N,H,W,Ci,Co,res[0],fl/res[0]/1e9,res[1],fl/res[1]/1e9,res[1]/res[0]), flush=True)
