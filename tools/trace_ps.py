#!/usr/bin/env python3
"""Phase trace of single conv_ps_k launches (pre-split operands, conv_ps.hip): private build with -DPMF_CONV_TRACE, thread 0
of every workgroup stamps s_memtime at phase boundaries.
usage: python tools/trace_ps.py [case-substring]      env: TRACE_CFG (tile cfg), TRACE_DEFS (extra -D switches)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pmf_amd import _lib as L
from tests import gpu_helpers as G
from tools.bench_conv import CASES
SLOTS = 128

def build():
    # conv_ps.hip alone is rebuilt (with the trace stamps and any -D switches); the host helpers it calls come from the
    # library's own conv_fwd.o (built by make, no trace)
    so, obj = "/tmp/libpmf_ps_trace.so", "/tmp/conv_ps_trace.o"
    csrc = os.path.join(ROOT, "pmf_amd/csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                           "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DPMF_CONV_TRACE",
                           "-DPMF_TRACE_SLOTS=%d" % SLOTS] + os.environ.get("TRACE_DEFS", "").split() +
                          [os.path.join(csrc, "conv_ps.hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj,
                           os.path.join(csrc, "conv_fwd.o"), "-o", so])
    return C.CDLL(so)

def main(filt):
    tl = build()
    tl.pmf_conv_fwd.argtypes = [C.c_void_p, C.c_void_p]
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    for name, N, H, W, ci, co, k, dil in CASES:
        if filt and filt not in name: continue
        pad = dil * (k - 1) // 2
        if k == 2: pad = 1
        x = torch.randn(N, H, W, ci, device="cuda"); w = torch.randn(co, ci, k, k) * 0.05
        ldw = (co + 63) // 64 * 64
        w3 = G.pack_fwd_s3(w, ci, ldw); out = torch.empty(N, H, W, co, device="cuda")
        xs = G.presplit(x, ci)
        d = G.conv_desc([dict(x=x, C=ci)], w3, ldw, None, out, N, H, W, co, G.taps_of(k, k, dil, pad), 1, 1)
        d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        d.cfg = int(os.environ.get("TRACE_CFG", "0"), 0)
        d.w, d.w_s3 = None, w3.data_ptr()
        d.src[0].xs = xs.data_ptr()
        buf = torch.zeros(16384 * SLOTS, dtype=torch.int64, device="cuda")
        s = G.stream()
        for _ in range(300): tl.pmf_conv_fwd(C.byref(d), s)
        tl.pmf_conv_ps_trace_set(C.c_void_p(buf.data_ptr()))
        tl.pmf_conv_fwd(C.byref(d), s)
        torch.cuda.synchronize()
        tl.pmf_conv_ps_trace_set(C.c_void_p(0))
        t = buf.cpu().numpy().reshape(-1, SLOTS)
        nwg = int((t[:, SLOTS - 1] > 0).sum()); t = t[:nwg]
        cnt = int(t[0, SLOTS - 1])
        st_ = t[:, 0]; en = t[:, cnt - 1]
        w0 = t[:, SLOTS - 5].astype(np.int64); w1 = t[:, SLOTS - 4].astype(np.int64); wb = w0.min()
        clk = np.median((en - st_) / np.maximum(w1 - w0, 1)) / 10.0
        print("== %s cfg %#x: %d workgroups, %d stamps; launch %.1f us wall; workgroup lifetime median %.1f us; starts median %.1f max %.1f us; clock %.2f GHz" % (
            name, d.cfg, nwg, cnt, (w1.max() - wb) / 100.0, np.median(w1 - w0) / 100.0, np.median(w0 - wb) / 100.0,
            (w0.max() - wb) / 100.0, clk))
        d_ = np.diff(t[:, :cnt], axis=1).astype(np.float64)
        med = np.median(d_, axis=0)
        nst = (cnt - 7) // 4
        print("   prologue: tables %d, first issue %d ticks" % (med[0], med[1]))
        a = med[2:2 + 4 * nst].reshape(nst, 4)
        print("   per stage (median over workgroups), ticks: wait+barrier top | MFMA half 0 | wait+barrier mid | MFMA half 1")
        if os.environ.get("TRACE_VERBOSE"):
            for i in range(nst):
                print("     stage %2d: %6d %6d %6d %6d" % (i, a[i, 0], a[i, 1], a[i, 2], a[i, 3]))
        print("   stage mean: top %.0f  h0 %.0f  mid %.0f  h1 %.0f  = %.0f ticks/stage" % (
            a[1:, 0].mean(), a[1:, 1].mean(), a[1:, 2].mean(), a[1:, 3].mean(), a[1:].sum(1).mean()))
        tail = med[2 + 4 * nst:]
        print("   tail:", " ".join("%d" % v for v in tail), " | total per workgroup median %d ticks" % np.median(en - st_))

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")
