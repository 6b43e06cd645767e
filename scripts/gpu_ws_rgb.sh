#!/bin/bash
# round 6: the specialised persistent direct-sum kernel with ToRGB in its epilogue (dconv_ws_rgb_kernel) on layer 18
out=gpurun_out/${1:-r06l}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "direct16_conv_matches_direct_fp32_kernel_and_oracle and specialised" > $out/pytest_ws_rgb.log 2>&1; tail -3 $out/pytest_ws_rgb.log
python - <<'PY' 2>&1 | tee $out/layer18_forms.jsonl
import json, math, os, torch
from rewriting_amd import hip
DEV='cuda:0'; B=64; cin=cout=32; res=1024
g=torch.Generator().manual_seed(0)
x=torch.randn(B,cin,res,res,device=DEV); wt=torch.randn(1,cout,cin,3,3,generator=g).to(DEV)
style=(1+0.3*torch.randn(B,cin,generator=g)).to(DEV); s=1/math.sqrt(cin*9)
dm=hip.demod(hip.weight_sqsum(wt,s),style); bias=torch.randn(cout,generator=g).to(DEV); nw=torch.tensor([0.1],device=DEV)
noise=torch.randn(B,res*res,device=DEV); amax=hip.absmax(x)
wrgb=torch.randn(3,cout,device=DEV); srgb=1+0.3*torch.randn(B,cout,device=DEV); brgb=torch.randn(3,device=DEV); skip=torch.randn(B,3,res,res,device=DEV)
pk=hip.pack_conv_weight_direct16(wt); uf=hip.pack_conv_weight_wino4(wt, split=True)
xs = x*style[:,:,None,None]
def timed(fn,it=5):
    fn(); torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
ep=dict(demod=dm,noise=noise,noise_w=nw,bias=bias,act=True)
rows={}
os.environ['RW_DCONV_V']='2'
rows['direct ws rgb (style on load)']=timed(lambda: hip.conv3x3_direct16_to_rgb(x,pk,cout,s,wrgb,srgb,brgb,skip,1/math.sqrt(cout),style=style,x_amax=amax,**ep))
os.environ['RW_DCONV_V']='1'
rows['direct one-role rgb (style on load)']=timed(lambda: hip.conv3x3_direct16_to_rgb(x,pk,cout,s,wrgb,srgb,brgb,skip,1/math.sqrt(cout),style=style,x_amax=amax,**ep))
del os.environ['RW_DCONV_V']
amaxs=hip.absmax(xs)
rows['direct one-role rgb (prescaled input)']=timed(lambda: hip.conv3x3_direct16_to_rgb(xs,pk,cout,s,wrgb,srgb,brgb,skip,1/math.sqrt(cout),x_amax=amaxs,**ep))
rows['F(4x4) split rgb (prescaled input)']=timed(lambda: hip.conv3x3_wino4_to_rgb(xs,uf,cout,s,wrgb,srgb,brgb,skip,1/math.sqrt(cout),x_amax=amaxs,**ep))
rows['F(4x4) split rgb (style on load)']=timed(lambda: hip.conv3x3_wino4_to_rgb(x,uf,cout,s,wrgb,srgb,brgb,skip,1/math.sqrt(cout),style=style,x_amax=amax,**ep))
print(json.dumps({k: round(v,3) for k,v in rows.items()}))
PY
