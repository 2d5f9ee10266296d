import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F, numpy as np
from pmf_amd import _lib as L
from pmf_amd.utils.detinit import det_tensor
from tests import gpu_helpers as G
lib = L.lib()
def run(N,H,W,cin,cout,k,dil,pad,stride=1,act=0,bias=True,stats=False, onehot=None):
    x = det_tensor("x", (N,cin,H,W)); w = det_tensor("w", (cout,cin,k,k), -0.3, 0.3); b = det_tensor("b",(cout,))
    if onehot is not None:
        x = torch.zeros_like(x); x[0, onehot, H//2, W//2] = 1.0
        b = torch.zeros_like(b)
    ref = F.conv2d(x, w, b if bias else None, stride=stride, padding=pad, dilation=dil)
    if act: ref = F.leaky_relu(ref, 0.01)
    OH,OW = ref.shape[2:]
    ldw = (cout+63)//64*64
    wpk = G.pack_fwd(w, cin, ldw)
    out = torch.zeros(N,OH,OW,(cout+7)//8*8, device="cuda")
    st = torch.zeros(2*cout, device="cuda") if stats else None
    d = G.conv_desc([dict(x=G.nhwc(x), C=cin)], wpk, ldw, b.cuda() if bias else None, out, N, OH, OW, cout, G.taps_of(k,k,dil,pad), stride, act, stats=st)
    rc = lib.pmf_conv_fwd(C.byref(d), G.stream()); torch.cuda.synchronize()
    got = G.from_nhwc(out, cout)
    err = (got-ref).abs().max().item()
    print("N%d %dx%d cin%d cout%d k%d d%d s%d act%d bias%d stats%d oh%s rc=%d  maxerr=%.3e  ref_absmax=%.3f" % (N,H,W,cin,cout,k,dil,stride,act,bias,stats,onehot,rc,err,ref.abs().max()))
    return got, ref
for cin in (8,16,32):
    for cout in (32,64):
        for k,dil,pad in ((1,1,0),(3,1,1)):
            run(1,8,32,cin,cout,k,dil,pad)
run(1,8,32,16,32,1,1,0,bias=False)
run(1,8,32,16,32,1,1,0,act=1)
run(1,8,32,16,32,1,1,0,stats=True)
run(2,16,64,32,32,3,1,1,act=1,stats=True)
run(2,16,64,32,32,3,1,1)
run(2,32,64,32,32,3,1,1)
run(4,64,64,32,32,3,1,1)
for oh in (0,3,4,8,12,15):
    got, ref = run(1,8,32,16,32,1,1,0,onehot=oh)
    print("   got nz channels:", got[0,:,4,16].nonzero().flatten().tolist()[:8], " got pixel nz:", (got.abs().sum(1)>0).nonzero().tolist()[:6])
    w = det_tensor("w", (32,16,1,1), -0.3, 0.3)
    # which input channel's weights does the output equal?
    g = got[0,:,4,16]
    for c in range(16):
        if torch.allclose(g, w[:,c,0,0], atol=1e-6): print("   output == weights of channel", c)
