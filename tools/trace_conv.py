#!/usr/bin/env python3
"""Phase trace of single conv_fwd_k launches: builds a private copy of the conv kernel with -DPMF_CONV_TRACE (thread 0
of every workgroup stamps s_memtime at phase boundaries) and prints where the time of one launch goes.
usage: python tools/trace_conv.py [case-substring]     (cases of tools/bench_conv.py)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pmf_amd import _lib as L
from tests import gpu_helpers as G
from tools.bench_conv import CASES

def build():
    if os.environ.get("TRACE_LIB"):          # a prebuilt private copy (experiments)
        return C.CDLL(os.environ["TRACE_LIB"])
    so = "/tmp/libpmf_conv_trace.so"
    src = os.path.join(ROOT, "pmf_amd/csrc/conv_fwd.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DPMF_CONV_TRACE", src, "-o", so])
    return C.CDLL(so)

def main(filt, with_stats):
    tl = build()
    tl.pmf_conv_fwd.argtypes = [C.c_void_p, C.c_void_p]
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    for name, N, H, W, ci, co, k, dil in CASES:
        if filt and filt not in name: continue
        pad = dil * (k - 1) // 2
        x = torch.randn(N, H, W, ci, device="cuda"); w = torch.randn(co, ci, k, k) * 0.05
        ldw = (co + 63) // 64 * 64
        wpk = G.pack_fwd(w, ci, ldw); out = torch.empty(N, H, W, co, device="cuda")
        d = G.conv_desc([dict(x=x, C=ci)], wpk, ldw, None, out, N, H, W, co, G.taps_of(k, k, dil, pad), 1, 1)
        d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        d.cfg = int(os.environ.get("TRACE_CFG", "0"), 0)
        if os.environ.get("TRACE_S3"):
            w3 = G.pack_fwd_s3(w, ci, ldw)
            d.w, d.w_s3 = None, w3.data_ptr()
        if with_stats:
            rows = L.lib().pmf_conv_fwd_stat_rows(C.byref(d))
            st = torch.zeros(rows * 2 * co, dtype=torch.float64, device="cuda")
            d.stats = st.data_ptr()
        buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")   # up to 16384 workgroups
        s = G.stream()
        for _ in range(int(os.environ.get("TRACE_WARM", "3"))): tl.pmf_conv_fwd(C.byref(d), s)
        if os.environ.get("TRACE_WARM") is None: torch.cuda.synchronize()
        tl.pmf_conv_trace_set(C.c_void_p(buf.data_ptr()))
        tl.pmf_conv_fwd(C.byref(d), s)
        torch.cuda.synchronize()
        tl.pmf_conv_trace_set(C.c_void_p(0))
        t = buf.cpu().numpy().reshape(-1, 64)
        nwg = int((t[:, 63] > 0).sum()); t = t[:nwg]
        cnt = int(t[0, 63])
        st_ = t[:, 0]; en = t[:, cnt - 1]
        t0 = st_.min()
        wall = (t[:, 60].max() - t[:, 60].min())   # 100 MHz ticks between first and last workgroup END
        span = en.max() - t0
        # s_memtime frequency: calibrate against wall clock over the END stamps
        f = (en.max() - en.min()) / max(wall, 1) * 100.0 if wall > 0 else float("nan")   # MHz
        print("== %s: %d workgroups, %d stamps each, launch span %d ticks (s_memtime ~%.0f MHz -> %.1f us)" %
              (name, nwg, cnt, span, f, span / f if f == f else 0))
        xcc = t[:, 61] & 0xf; cu = (t[:, 62] >> 8) & 0xf; se = (t[:, 62] >> 13) & 0x7; sh = (t[:, 62] >> 12) & 1
        print("   start offset: median %d, p90 %d, max %d ticks;  end offset: min %d median %d" %
              (np.median(st_ - t0), np.percentile(st_ - t0, 90), (st_ - t0).max(), (en - t0).min(), np.median(en - t0)))
        w0 = t[:, 59].astype(np.int64); w1 = t[:, 60].astype(np.int64); wb = w0.min()
        print("   shader clock during this launch: %.2f GHz" % (np.median((en - st_) / np.maximum(w1 - w0, 1)) / 10.0))
        print("   wall clock (100 MHz): launch %.1f us; workgroup lifetime median %.1f us; starts: median %.1f us, max %.1f us" % (
            (w1.max() - wb) / 100.0, np.median(w1 - w0) / 100.0, np.median(w0 - wb) / 100.0, (w0.max() - wb) / 100.0))
        key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
        from collections import defaultdict
        per = defaultdict(list)
        for i, k_ in enumerate(key): per[k_].append((int(w0[i] - wb), int(w1[i] - wb)))
        conc = []
        for k_, iv in per.items():
            ev = sorted([(a, 1) for a, b in iv] + [(b, -1) for a, b in iv]); c = m = 0
            for _, d_ in ev: c += d_; m = max(m, c)
            conc.append(m)
        print("   workgroups per CU: min %d max %d; peak concurrent per CU: min %d median %d max %d" % (
            min(len(v) for v in per.values()), max(len(v) for v in per.values()), min(conc), int(np.median(conc)), max(conc)))
        print("   distinct (xcc,se,sh,cu): %d" % len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))))
        d_ = np.diff(t[:, :cnt], axis=1)
        med = np.median(d_, axis=0); p90 = np.percentile(d_, 90, axis=0)
        labels = ["acc zero + slot table (gA)", "tap / weight offsets", "settle + head (stage scalars)",
                  "first loads + DMA issue"]
        # stamps: start, 3 prologue stamps, after-first-issue, [X, stored+Y, half0+Z, half1]*nch, epilogue-start, stored, end
        nch = (cnt - 8) // 4
        for c in range(nch): labels += ["c%d barrier X" % c, "c%d store A + barrier Y" % c, "c%d mfma h0 (+loads) + barrier Z" % c, "c%d mfma h1" % c]
        labels += ["(to epilogue)", "epilogue stores", "stats reduce"]
        for i in range(cnt - 1):
            print("   %-34s median %7d  p90 %7d ticks" % (labels[i] if i < len(labels) else "?", med[i], p90[i]))
        print("   per-workgroup total: median %d ticks" % np.median(en - st_))

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "", True)
