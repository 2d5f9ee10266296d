import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pmf_amd import _lib as L
from tests import gpu_helpers as G
lib = L.lib()
torch.manual_seed(0)
for (N, H, W, k, dil, cin_real, co) in ((2, 32, 1024, 7, 1, 3, 64), (2, 64, 2048, 7, 1, 3, 64), (2, 64, 512, 7, 1, 3, 64), (2, 64, 2048, 3, 1, 5, 32),
                                         (2, 64, 2048, 3, 2, 5, 32), (1, 19, 70, 7, 1, 3, 64), (2, 32, 160, 3, 1, 5, 32)):
    pad = dil * (k - 1) // 2
    x = torch.zeros(N, 8, H, W); x[:, :cin_real] = torch.rand(N, cin_real, H, W)
    w = torch.zeros(co, 8, k, k); w[:, :cin_real] = torch.randn(co, cin_real, k, k) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, padding=pad, dilation=dil)
    ldw = (co + 63) // 64 * 64
    xs = G.nhwc(x)
    taps = G.taps_of(k, k, dil, pad)
    for cfg in (0, 32 | (1 << 8) | (1 << 16), 32 | (2 << 8) | (1 << 16), 64 | (1 << 8) | (1 << 16), 64 | (2 << 8) | (1 << 16)):
        if (cfg & 255) == 64 and co <= 32: continue
        res = {}
        for kind in ("f32", "s3"):
            out = torch.zeros(N, H, W, (co + 7) // 8 * 8, device="cuda")
            wpk = G.pack_fwd(w, 8, ldw) if kind == "f32" else G.pack_fwd_s3_stem(w, ldw)
            d = G.conv_desc([dict(x=xs, C=8)], wpk, ldw, None, out, N, H, W, co, taps, 1, 0)
            d.cfg = cfg
            if kind == "s3": d.w, d.w_s3 = None, wpk.data_ptr()
            rc = lib.pmf_conv_fwd(C.byref(d), G.stream()); torch.cuda.synchronize()
            got = G.from_nhwc(out, co).double()
            e = (got - ref).abs()
            res[kind] = (rc, float(e.max() / ref.abs().max()), int((e > 1e-4 * ref.abs().max()).sum()))
        print("%dx%dx%d k%d d%d cin%d->%d cfg %#x  f32 %s  s3 %s" % (N, H, W, k, dil, cin_real, co, cfg, res["f32"], res["s3"]), flush=True)
