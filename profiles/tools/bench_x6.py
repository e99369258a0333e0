import os, sys, torch, math
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','golden'))
from free_hunch_amd import _lib as L
from test_hip_unet import wino_weights
lib=L.load(); dev=torch.device('cuda:0')
def split_w(wd):
    h=wd.to(torch.bfloat16); r=wd-h.float(); m=r.to(torch.bfloat16); r=r-m.float(); l=r.to(torch.bfloat16)
    Co,T,Ci=wd.shape
    return torch.stack([h,m,l]).reshape(3,Co,T,Ci//32,32).permute(0,2,3,1,4).contiguous()
# accuracy vs float64
torch.manual_seed(0)
for (N,H,W,Ci,Co) in [(1,32,32,128,128),(2,16,16,512,256)]:
    x=torch.randn(N,H,W,Ci,device=dev)*torch.rand(N,H,W,Ci,device=dev)*3; w=torch.randn(Co,Ci,3,3,device=dev)*0.05; b=torch.randn(Co,device=dev)
    ref=torch.nn.functional.conv2d(x.permute(0,3,1,2).double(), w.double(), b.double(), padding=1).permute(0,2,3,1)
    wd=w.permute(0,2,3,1).reshape(Co,9,Ci).contiguous(); wx=split_w(wd); wu=wino_weights(w)
    out=torch.empty(N,H,W,Co,device=dev)
    errs={}
    L.check(lib.fh_conv2d_nhwc(x.data_ptr(),wd.data_ptr(),b.data_ptr(),None,out.data_ptr(),None,1,N,H,W,Ci,Co,3,3,1,1,L.stream()),"c"); errs['f32 mfma']=float((out.double()-ref).abs().max())
    L.check(lib.fh_conv3x3_wino_nhwc(x.data_ptr(),wu.data_ptr(),b.data_ptr(),None,out.data_ptr(),N,H,W,Ci,Co,L.stream()),"w"); errs['wino f32']=float((out.double()-ref).abs().max())
    out.zero_()
    L.check(lib.fh_conv2d_x6_nhwc(x.data_ptr(),wx.data_ptr(),b.data_ptr(),None,out.data_ptr(),None,1,N,H,W,Ci,Co,3,3,1,1,L.stream()),"x"); errs['bf16x6']=float((out.double()-ref).abs().max())
    t32=torch.nn.functional.conv2d(x.permute(0,3,1,2), w, b, padding=1).permute(0,2,3,1); errs['torch f32 (MIOpen)']=float((t32.double()-ref).abs().max())
    print((N,H,W,Ci,Co), "ref max %.2f"%float(ref.abs().max()), {k:"%.2e"%v for k,v in errs.items()}, flush=True)
shapes=[(8,256,256,128,128),(8,256,256,256,128),(8,128,128,256,256),(8,64,64,256,256),(1,256,256,128,128),(8,32,32,256,256),(8,16,16,512,512),(8,8,8,512,512)]
for (N,H,W,Ci,Co) in shapes:
    x=torch.randn(N,H,W,Ci,device=dev); w=torch.randn(Co,Ci,3,3,device=dev)*0.03; b=torch.zeros(Co,device=dev); out=torch.empty(N,H,W,Co,device=dev)
    wd=w.permute(0,2,3,1).reshape(Co,9,Ci).contiguous(); wx=split_w(wd)
    ks=lib.fh_conv2d_splitk(N,H,W,Ci,Co,3,3); ws=torch.empty(max(ks,1),N*H*W,Co,device=dev)
    f1=lambda: L.check(lib.fh_conv2d_x6_nhwc(x.data_ptr(),wx.data_ptr(),b.data_ptr(),None,out.data_ptr(),ws.data_ptr(),ks,N,H,W,Ci,Co,3,3,1,1,L.stream()),"x")
    f2=lambda: L.check(lib.fh_conv2d_nhwc(x.data_ptr(),wd.data_ptr(),b.data_ptr(),None,out.data_ptr(),ws.data_ptr(),ks,N,H,W,Ci,Co,3,3,1,1,L.stream()),"c")
    res=[]
    for f in (f1,f2):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/10)
    fl=2.0*N*H*W*Ci*Co*9
    print("N%d %3dx%-3d Ci%4d Co%4d ks%d: x6 %7.3f ms (%5.1f fp32-equiv TF/s)   direct f32 %7.3f ms (%5.1f TF/s)  speedup %.2f" % (N,H,W,Ci,Co,ks,res[0],fl/res[0]/1e9,res[1],fl/res[1]/1e9,res[1]/res[0]), flush=True)
