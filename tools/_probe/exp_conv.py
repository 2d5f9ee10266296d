import ctypes as C, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import torch
from pmf_amd import _lib as L
from tests import gpu_helpers as G
from tools.bench_conv import CASES, timeit
def main(filt, cfg, variants):
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    for name, N, H, W, ci, co, k, dil in CASES:
        if filt != name: continue
        pad = 1 if k == 2 else dil * (k - 1) // 2
        x = torch.randn(N, H, W, ci, device="cuda"); w = torch.randn(co, ci, k, k) * 0.05
        ldw = (co + 63) // 64 * 64
        w3 = G.pack_fwd_s3(w, ci, ldw); out = torch.empty(N, H, W, co, device="cuda")
        taps = G.taps_of(k, k, dil, pad)
        gf = 2.0 * N * H * W * ci * co * k * k / 1e9
        for v in variants:
            tl = C.CDLL(os.path.join(ROOT, "tools/_probe/libcf_%s.so" % v))
            tl.pmf_conv_fwd.argtypes = [C.c_void_p, C.c_void_p]
            d = G.conv_desc([dict(x=x, C=ci)], w3, ldw, None, out, N, H, W, co, taps, 1, 1)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
            d.cfg = cfg
            d.w, d.w_s3 = None, w3.data_ptr()
            st = G.stream()
            us = timeit(lambda: tl.pmf_conv_fwd(C.byref(d), st))
            print("%-22s cfg %#8x %-10s %7.1f us %6.1f TF/s" % (name, cfg, v, us, gf / us * 1e3), flush=True)
main(sys.argv[1], int(sys.argv[2], 0), sys.argv[3:])
