import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from luminoth_amd import kernels as K
from oracle import torch_ops as ot
K.WINOGRAD = False
dev='cuda:0'
rs=np.random.RandomState(0)
for compute in ('f16','bf16'):
  for (N,H,W,C,Kc,R,stride) in ((2,32,32,256,256,3,1),(2,32,32,256,1024,1,1),(1,40,40,128,128,3,2)):
    x=np.maximum(rs.randn(N,H,W,C),0).astype(np.float32); w=(rs.randn(R,R,C,Kc)*np.sqrt(2/(R*R*C))).astype(np.float32)
    pad='SAME' if stride==1 else 'SAME_EXPLICIT'
    d=K.conv_desc(x.shape,w.shape,stride,1,pad,None,compute)
    y=K.conv2d_fwd(d,torch.tensor(x).to(dev),torch.tensor(w).to(dev)).cpu()
    xt=torch.tensor(x,requires_grad=True); wt=torch.tensor(w,requires_grad=True)
    yo=ot.conv2d_nhwc(xt,wt,stride,1,pad,quant=compute)
    g=(rs.randn(*yo.shape)*1e-3).astype(np.float32)
    yo.backward(torch.tensor(g))
    dx=K.conv2d_bwd_data(d,torch.tensor(g).to(dev),torch.tensor(w).to(dev)).cpu()
    dw=K.conv2d_bwd_weight(d,torch.tensor(x).to(dev),torch.tensor(g).to(dev)).cpu()
    f=lambda a,b: float((a-b).abs().max()/b.abs().max())
    print(compute,(N,H,W,C,Kc,R,stride),'fwd %.2e bwd_data %.2e bwd_weight %.2e'%(f(y,yo.detach()),f(dx,xt.grad),f(dw.reshape(wt.grad.shape),wt.grad)))
