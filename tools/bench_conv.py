#!/usr/bin/env python3
"""Micro-benchmark of single conv launches through the C ABI (forward kernel and weight gradient).
usage: python tools/bench_conv.py [fwd|wgrad|all] [case-substring]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmf_amd import _lib as L
from tests import gpu_helpers as G
lib = L.lib()
CASES = [  # name, N, H, W, Cin, Cout, k, dil
    ("full_stem_8_64_7x7", 2, 64, 2048, 8, 64, 7, 1),
    ("full_32_32_3x3", 2, 64, 2048, 32, 32, 3, 1),
    ("full_32_32_3x3d2", 2, 64, 2048, 32, 32, 3, 2),
    ("full_64_64_3x3d2", 2, 64, 2048, 64, 64, 3, 2),
    ("full_32_32_1x1", 2, 64, 2048, 32, 32, 1, 1),
    ("full_192_64_1x1", 2, 64, 2048, 192, 64, 1, 1),
    ("full_64_64_1x1", 2, 64, 2048, 64, 64, 1, 1),
    ("full_64_192_1x1", 2, 64, 2048, 64, 192, 1, 1),
    ("half_128_384_1x1", 2, 32, 1024, 128, 384, 1, 1),
    ("half_384_128_1x1", 2, 32, 1024, 384, 128, 1, 1),
    ("quar_768_256_1x1", 2, 16, 512, 768, 256, 1, 1),
    ("full_64_64_2x2d2", 2, 64, 2048, 64, 64, 2, 2),
    ("half_128_128_2x2d2", 2, 32, 1024, 128, 128, 2, 2),
    ("full_32_32_2x2d2", 2, 64, 2048, 32, 32, 2, 2),
    ("quar_256_256_2x2d2", 2, 16, 512, 256, 256, 2, 2),
    ("8th_256_256_2x2d2", 2, 8, 256, 256, 256, 2, 2),
    ("half_64_64_3x3", 2, 32, 1024, 64, 64, 3, 1),
    ("half_128_128_3x3d2", 2, 32, 1024, 128, 128, 3, 2),
    ("quar_128_128_3x3", 2, 16, 512, 128, 128, 3, 1),
    ("quar_256_256_3x3d2", 2, 16, 512, 256, 256, 3, 2),
    ("8th_256_256_3x3", 2, 8, 256, 256, 256, 3, 1),
    ("16th_512_512_3x3", 2, 4, 128, 512, 512, 3, 1),
    ("16th_1024_256_1x1", 2, 4, 128, 1024, 256, 1, 1),      # ResNet-50 bottlenecks: split-K launches + conv_finish_k
    ("32nd_2048_512_1x1", 2, 2, 64, 2048, 512, 1, 1),
    ("32nd_512_2048_1x1", 2, 2, 64, 512, 2048, 1, 1),
]
def timeit(fn, n=200):
    # the shader clock needs tens of milliseconds of load to ramp from idle: warm up long enough, then time
    for _ in range(400): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run(which, filt):
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    for name, N, H, W, ci, co, k, dil in CASES:
        if filt and filt not in name: continue
        pad = dil * (k - 1) // 2
        if k == 2: pad = 1
        x = torch.randn(N, H, W, ci, device="cuda"); w = torch.randn(co, ci, k, k) * 0.05
        ldw = (co + 63) // 64 * 64
        wpk = G.pack_fwd(w, ci, ldw); out = torch.empty(N, H, W, co, device="cuda")
        taps = G.taps_of(k, k, dil, pad)
        gf = 2.0 * N * H * W * ci * co * k * k / 1e9
        if which in ("fwd", "all"):
            d = G.conv_desc([dict(x=x, C=ci)], wpk, ldw, None, out, N, H, W, co, taps, 1, 1)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
            st = G.stream()
            us = timeit(lambda: lib.pmf_conv_fwd(C.byref(d), st))
            print("fwd   %-22s %8.1f us  %6.1f TF/s" % (name, us, gf / us * 1e3), flush=True)
        if which == "fwdst":   # with BatchNorm statistics rows (the split-K tail then writes them)
            d = G.conv_desc([dict(x=x, C=ci)], wpk, ldw, None, out, N, H, W, co, taps, 1, 0)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
            rows = lib.pmf_conv_fwd_stat_rows(C.byref(d))
            stt = torch.empty(rows * 2 * co, dtype=torch.float64, device="cuda"); d.stats = stt.data_ptr()
            st = G.stream()
            us = timeit(lambda: lib.pmf_conv_fwd(C.byref(d), st))
            print("fwdst %-22s %8.1f us  %6.1f TF/s  (%d stat rows)" % (name, us, gf / us * 1e3, rows), flush=True)
        if which == "s3":      # split-bf16 path, every tile configuration, next to the fp32 MFMA path
            w3 = G.pack_fwd_s3_stem(w, ldw) if (k == 7 and ci == 8) else G.pack_fwd_s3(w, ci, ldw)
            cfgs = (0, 32 | (1 << 8) | (1 << 16), 64 | (1 << 8) | (1 << 16), 32 | (2 << 8) | (1 << 16), 64 | (2 << 8) | (1 << 16))
            if os.environ.get("S3_CFGS"): cfgs = tuple(int(x, 0) for x in os.environ["S3_CFGS"].split(","))
            for cfg in cfgs:
                if (cfg & 0xff) == 64 and co <= 32: continue
                res = []
                for kind in ("f32", "s3"):
                    d = G.conv_desc([dict(x=x, C=ci)], wpk, ldw, None, out, N, H, W, co, taps, 1, 1)
                    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
                    d.cfg = cfg
                    if kind == "s3": d.w, d.w_s3 = None, w3.data_ptr()
                    st = G.stream()
                    res.append(timeit(lambda: lib.pmf_conv_fwd(C.byref(d), st)))
                print("%-22s cfg %#8x  f32 %7.1f us %6.1f TF/s | s3 %7.1f us %6.1f TF/s" % (
                    name, cfg, res[0], gf / res[0] * 1e3, res[1], gf / res[1] * 1e3), flush=True)
        if which in ("wgrad", "all"):
            dz = torch.randn(N, H, W, co, device="cuda")
            wd = L.WgradDesc()
            wd.N, wd.OH, wd.OW, wd.Cout, wd.nsrc = N, H, W, co, 1
            wd.src[0].x, wd.src[0].C, wd.src[0].ldc, wd.src[0].H, wd.src[0].W = x.data_ptr(), ci, ci, H, W
            wd.ntaps = len(taps)
            for i, (dy, dx) in enumerate(taps): wd.tdy[i], wd.tdx[i], wd.tap_widx[i] = dy, dx, i
            wd.in_stride = 1; wd.dz, wd.dz_ldc = dz.data_ptr(), co
            wd.flags = L.WGRAD_S3 if os.environ.get("WG_S3", "1") != "0" else 0     # split-bf16 kernels (the plan's default)
            wd.Cin_real, wd.KHW = ci, k * k          # (before the split count: the few-channel stem kernel keys on it)
            wd.nsplit = 1; wd.nsplit = lib.pmf_conv_wgrad_nsplit(C.byref(wd))
            part = torch.empty(lib.pmf_conv_wgrad_workspace(C.byref(wd)), dtype=torch.uint8, device="cuda")
            gw = torch.empty(co, ci, k, k, device="cuda")
            wd.partial, wd.dw_oihw, wd.Cin_real, wd.KHW = part.data_ptr(), gw.data_ptr(), ci, k * k
            st = G.stream()
            us = timeit(lambda: lib.pmf_conv_wgrad(C.byref(wd), st))
            print("wgrad %-22s %8.1f us  %6.1f TF/s  (nsplit %d)" % (name, us, gf / us * 1e3, wd.nsplit), flush=True)
if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "all", sys.argv[2] if len(sys.argv) > 2 else "")
