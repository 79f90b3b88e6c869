import sys, os, math, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
from mivos_amd import ops, _lib
from mivos_amd._lib import ConvDesc, check
from mivos_amd.ops import ConvLayer
torch.set_grad_enabled(False)
DEV="cuda:0"; lib=_lib.load(); st=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
n,h,w,cin,cout,k,s=5,120,216,256,256,3,1
x=torch.randn(n,h,w,cin,device=DEV)
L=ConvLayer.pack(torch.randn(cout,cin,k,k)*0.05, torch.randn(cout)*0.1, None, s, 1).to(DEV)
xs=torch.zeros(n,h+2,w+2,cin,device=DEV); xin=xs[:,1:h+1,1:w+1]
rs,ns=(w+2)*cin,(h+2)*(w+2)*cin
check(lib.mivos_pack_activation_sh32(x.data_ptr(), h*w*cin, w*cin, cin, xin.data_ptr(), ns, rs, cin, n, h, w, cin, 0, st()))
sh=14-math.floor(math.log2(float(L.w.abs().max())))
wd=torch.empty(lib.mivos_pack_weights_f16x3_dma_bytes(cout,k,k,cin),dtype=torch.uint8,device=DEV)
check(lib.mivos_pack_weights_f16x3_dma(L.w.data_ptr(), wd.data_ptr(), cout,k,k,cin, 2.0**sh, st()))
_,sc=L.f16x3()
y=torch.empty(n,h,w,cout,device=DEV)
ws=torch.zeros(1<<20,dtype=torch.uint8,device=DEV)
d=ConvDesc(); d.x,d.w,d.scale,d.bias,d.y=xin.data_ptr(),wd.data_ptr(),sc.data_ptr(),L.bias.data_ptr(),y.data_ptr()
d.N,d.H,d.W,d.Cin,d.Cout,d.KH,d.KW=n,h,w,cin,cout,k,k; d.stride,d.pad,d.Ho,d.Wo,d.split=1,1,h,w,cout
d.relu_in,d.relu_out,d.precision=0,1,2; d.x_nstride,d.x_rstride,d.x_pstride,d.x_border,d.x_format=ns,rs,cin,1,1; d.y_nstride,d.y_pstride=h*w*cout,cout
d.workspace,d.workspace_bytes=ws.data_ptr(),ws.numel()
for _ in range(3): check(lib.mivos_conv2d_fused(C.byref(d), st()))
torch.cuda.synchronize()
t=ws.view(torch.int64)[:1024].cpu().view(8,128)
names=["L0 start","L0 done","M0 start","M0 done","L1 start","L1 done","M1 start","M1 done"]
for wv in (0,4):
    r=t[wv].tolist()
    for step in range(5):
        seg=r[step*8:(step+1)*8]
        print("wave",wv,"step",12+step,"L0 %d | bar %d | M0 %d | bar %d | L1 %d | bar %d | M1 %d | bar+next %d" % tuple([seg[i+1]-seg[i] for i in range(7)]+[r[(step+1)*8]-seg[7]]))
