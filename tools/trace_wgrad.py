#!/usr/bin/env python3
"""Phase trace of one conv_wgrad_s3_k launch: builds a private copy of conv_wgrad.hip with -DPMF_WG_TRACE (thread 0 of
every workgroup stamps s_memtime at phase boundaries: start, loop entry, then per tile X-barrier / input tile stored /
Y-barrier / B fragments ready / MFMAs done, loop exit, end) and prints the median cycle count of every phase.
usage: python tools/trace_wgrad.py [case-substring] [nsplit]     (cases of tools/bench_conv.py)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pmf_amd import _lib as L
from tests import gpu_helpers as G
from tools.bench_conv import CASES

def build():
    so = "/tmp/libpmf_wg_trace.so"
    src = os.path.join(ROOT, "pmf_amd/csrc/conv_wgrad.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-munsafe-fp-atomics", "-DPMF_WG_TRACE"] +        # (as the Makefile: no -amdgpu-mfma-vgpr-form for this file)
                          os.environ.get("TRACE_DEFS", "").split() + [src, "-o", so])     # e.g. TRACE_DEFS="-DPMF_WG_NOMFMA"
    return C.CDLL(so)

def main(filt, ns):
    tl = build()
    tl.pmf_conv_wgrad.argtypes = [C.c_void_p, C.c_void_p]
    lib = L.lib()
    for name, N, H, W, ci, co, k, dil in CASES:
        if filt and filt not in name: continue
        pad = dil * (k - 1) // 2
        if k == 2: pad = 1
        x = torch.randn(N, H, W, ci, device="cuda"); dz = torch.randn(N, H, W, co, device="cuda")
        taps = G.taps_of(k, k, dil, pad)
        wd = L.WgradDesc()
        wd.N, wd.OH, wd.OW, wd.Cout, wd.nsrc = N, H, W, co, 1
        wd.src[0].x, wd.src[0].C, wd.src[0].ldc, wd.src[0].H, wd.src[0].W = x.data_ptr(), ci, ci, H, W
        wd.ntaps = len(taps)
        for i, (dy, dx) in enumerate(taps): wd.tdy[i], wd.tdx[i], wd.tap_widx[i] = dy, dx, i
        wd.in_stride = 1; wd.dz, wd.dz_ldc = dz.data_ptr(), co
        wd.flags = L.WGRAD_S3
        wd.Cin_real, wd.KHW = ci, k * k
        wd.nsplit = 1; wd.nsplit = ns if ns else lib.pmf_conv_wgrad_nsplit(C.byref(wd))
        part = torch.empty(lib.pmf_conv_wgrad_workspace(C.byref(wd)), dtype=torch.uint8, device="cuda")
        gw = torch.empty(co, ci, k, k, device="cuda")
        wd.partial, wd.dw_oihw, wd.Cin_real, wd.KHW = part.data_ptr(), gw.data_ptr(), ci, k * k
        st = G.stream()
        for _ in range(300): tl.pmf_conv_wgrad(C.byref(wd), st)
        torch.cuda.synchronize()
        buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
        tl.pmf_wg_trace_set(C.c_void_p(buf.data_ptr()))
        tl.pmf_conv_wgrad(C.byref(wd), st)
        torch.cuda.synchronize()
        tl.pmf_wg_trace_set(C.c_void_p(0))
        t = buf.cpu().numpy().reshape(-1, 64)
        t = t[t[:, 63] > 0]
        cnt = int(np.median(t[:, 63]))
        t = t[t[:, 63] == cnt]
        d = np.diff(t[:, :cnt].astype(np.int64), axis=1)
        span = int((t[:, cnt - 1].max() - t[:, 0].min()))
        print("== %s nsplit %d: %d workgroups, %d stamps, launch span %d ticks; per-workgroup total median %d" % (
            name, wd.nsplit, len(t), cnt, span, int(np.median(t[:, cnt - 1] - t[:, 0]))))
        names = ["prologue"]
        per = ["X barrier", "split+store X", "Y barrier", "fetch + dz wait + B prep", "108 MFMAs"]
        if os.environ.get("PMF_WG_S3N", "4") not in ("0", "1") and co % 64 == 0 and os.environ.get("PMF_WG_SWP", "1") != "0" \
                and os.environ.get("PMF_WG_W8", "0") != "1":
            per = ["barrier + loads landed", "MFMAs + split(t+1) + dz prep", "fetch(t+2)"]     # N-split body: three stamps per tile
        elif os.environ.get("PMF_WG_SWP", "1") != "0":     # software-pipelined body: three stamps per tile
            per = ["barrier", "dz wait + B prep", "108 MFMAs + split(t+1)", "fetch(t+2)"]
            if os.environ.get("PMF_WG_W8", "0") == "1":
                per = ["barrier", "dz DMA + input wait", "MFMAs + split(t+1) + fetch(t+2) + B prep(t+1)", "dz wait"]
        ntile = (cnt - 1 - 1 - 2) // len(per)
        for i in range(ntile): names += ["t%d %s" % (i, p) for p in per]
        names += ["loop exit", "reduce+write"]
        med = np.median(d, axis=0)
        for n_, m in zip(names, med): print("   %-26s %8d" % (n_, m))
        agg = {p: 0 for p in per}
        for i in range(1, ntile):
            for j, p in enumerate(per): agg[p] += med[1 + len(per) * i + j]
        print("   per tile (tiles 1..%d):" % (ntile - 1), {p: int(v / max(ntile - 1, 1)) for p, v in agg.items()})

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "quar_256_256_3x3d2", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
