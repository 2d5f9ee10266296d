#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound BatchNorm-backward kernels through the C ABI: effective bandwidth (algorithmic bytes /
time) of pmf_bn_bwd_reduce (2 reads) and pmf_bn_bwd_apply (2 reads + 1 write) at the shapes of the network.
usage: python tools/bench_elem.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmf_amd import _lib as L
lib = L.lib()
P = lambda t: C.c_void_p(t.data_ptr())
def timeit(fn, n=100):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for N, H, W, Cc in ((2, 64, 2048, 32), (2, 64, 2048, 64), (2, 32, 1024, 128), (2, 16, 512, 256), (2, 8, 256, 256)):
    npix = N * H * W
    # NSET rotating buffer sets: more than the 256 MB memory-side cache, so every pass streams from HBM
    NSET = max(1, int(os.environ.get("NSET", "8")))
    gys = [torch.randn(npix, Cc, device="cuda") for _ in range(NSET)]
    as_ = [torch.randn(npix, Cc, device="cuda") for _ in range(NSET)]
    dzs = [torch.empty(npix, Cc, device="cuda") for _ in range(NSET)]
    it = [0]
    def nxt():
        it[0] = (it[0] + 1) % NSET
        return gys[it[0]], as_[it[0]], dzs[it[0]]
    rows = lib.pmf_col_rows(C.c_int64(npix), C.c_int32(Cc))
    part = torch.zeros(rows * 2 * Cc, dtype=torch.float64, device="cuda")
    coef = torch.zeros(3 * Cc, device="cuda"); mean = torch.zeros(Cc, device="cuda"); istd = torch.ones(Cc, device="cuda")
    gam = torch.ones(Cc, device="cuda"); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    dbr = torch.zeros(rows * Cc, device="cuda")
    def f_r():
        gy, a, dz = nxt()
        lib.pmf_bn_bwd_reduce(P(gy), C.c_int32(Cc), P(a), C.c_int32(Cc), C.c_int64(npix), C.c_int32(Cc), P(mean),
                              P(gam), P(istd), C.c_int32(1), P(part), P(coef), P(dg), P(db), st)
    def f_a():
        gy, a, dz = nxt()
        lib.pmf_bn_bwd_apply(P(gy), C.c_int32(Cc), P(a), C.c_int32(Cc), C.c_int64(npix), C.c_int32(Cc), P(coef),
                             P(mean), C.c_int32(1), P(dz), C.c_int32(Cc), P(dbr), C.c_int32(Cc), st)
    t_r, t_a = timeit(f_r), timeit(f_a)
    mb = npix * Cc * 4 / 1e6
    print("%dx%dx%dx%-4d rows %4d  reduce+fold %6.1f us %5.2f TB/s | apply %6.1f us %5.2f TB/s" % (
        N, H, W, Cc, rows, t_r, 2 * mb / t_r, t_a, 3 * mb / t_a), flush=True)
