#!/usr/bin/env python3
"""Launch ONE bench_conv case a few times through the C ABI (for rocprofv3 --pmc runs).
usage: run_one_conv.py case cfg [s3|f32] [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmf_amd import _lib as L
from tests import gpu_helpers as G
from tools.bench_conv import CASES
lib = L.lib()
filt, cfg = sys.argv[1], int(sys.argv[2], 0)
kind = sys.argv[3] if len(sys.argv) > 3 else "s3"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
for name, N, H, W, ci, co, k, dil in CASES:
    if filt != name: continue
    pad = 1 if k == 2 else dil * (k - 1) // 2
    x = torch.randn(N, H, W, ci, device="cuda"); w = torch.randn(co, ci, k, k) * 0.05
    ldw = (co + 63) // 64 * 64
    wpk = G.pack_fwd(w, ci, ldw); w3 = G.pack_fwd_s3(w, ci, ldw); out = torch.empty(N, H, W, co, device="cuda")
    d = G.conv_desc([dict(x=x, C=ci)], wpk, ldw, None, out, N, H, W, co, G.taps_of(k, k, dil, pad), 1, 1)
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
    d.cfg = cfg
    if kind == "s3": d.w, d.w_s3 = None, w3.data_ptr()
    st = G.stream()
    for _ in range(reps): lib.pmf_conv_fwd(C.byref(d), st)
    torch.cuda.synchronize()
