#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound BatchNorm-backward kernels through the C ABI: effective bandwidth (algorithmic bytes /
time) of pmf_bn_bwd_reduce (2 reads) and pmf_bn_bwd_apply (2 reads + 1 write) at the shapes of the network, swept over the
launch shape (workgroup cap x pixels per trip, pmf_debug_col), and the one-launch small-map form (pmf_bn_bwd_small) against
the three-launch form.
usage: python tools/bench_elem.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmf_amd import _lib as L
lib = L.lib()
lib.pmf_debug_col.restype = C.c_int
lib.pmf_debug_col.argtypes = [C.c_int32, C.c_int32]
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
# warm the clock
x = torch.randn(64 << 20, device="cuda")
for _ in range(200): x.mul_(1.0001)
torch.cuda.synchronize()
SWEEP = [(512, 4), (1024, 4), (2048, 4), (512, 8), (1024, 8), (2048, 8)]
for N, H, W, Cc in ((2, 64, 2048, 32), (2, 64, 2048, 64), (2, 32, 1024, 128), (2, 32, 1024, 64), (2, 16, 512, 256), (2, 16, 512, 128),
                    (2, 8, 256, 256), (2, 4, 128, 256), (2, 4, 128, 512)):
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
    part = torch.zeros(4096 * 2 * Cc, dtype=torch.float64, device="cuda")
    coef = torch.zeros(3 * Cc, device="cuda"); mean = torch.zeros(Cc, device="cuda"); istd = torch.ones(Cc, device="cuda")
    gam = torch.ones(Cc, device="cuda"); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    dbr = torch.zeros(4096 * Cc, device="cuda")
    def f_r():
        gy, a, dz = nxt()
        lib.pmf_bn_bwd_reduce(P(gy), C.c_int32(Cc), P(a), C.c_int32(Cc), C.c_int64(npix), C.c_int32(Cc), P(mean),
                              P(gam), P(istd), C.c_int32(1), P(part), P(coef), P(dg), P(db), st)
    def f_a():
        gy, a, dz = nxt()
        lib.pmf_bn_bwd_apply(P(gy), C.c_int32(Cc), P(a), C.c_int32(Cc), C.c_int64(npix), C.c_int32(Cc), P(coef),
                             P(mean), C.c_int32(1), P(dz), C.c_int32(Cc), P(dbr), C.c_int32(Cc), st)
    def f_s():
        gy, a, dz = nxt()
        lib.pmf_bn_bwd_small(P(gy), C.c_int32(Cc), P(a), C.c_int32(Cc), C.c_int64(npix), C.c_int32(Cc), P(mean), P(gam),
                             P(istd), C.c_int32(1), C.c_int32(1), P(dz), C.c_int32(Cc), P(dbr), P(dg), P(db), st)
    mb = npix * Cc * 4 / 1e6
    line = "%dx%dx%dx%-4d (%5.1f MB)" % (N, H, W, Cc, mb)
    for cap, un in SWEEP:
        lib.pmf_debug_col(cap, un)
        rows = lib.pmf_col_rows(C.c_int64(npix), C.c_int32(Cc))
        t_r, t_a = timeit(f_r), timeit(f_a)
        line += " | cap %4d u%d rows %4d: red+fold %6.1f us %4.2f TB/s, apply %6.1f us %4.2f TB/s" % (
            cap, un, rows, t_r, 2 * mb / t_r, t_a, 3 * mb / t_a)
        if npix <= 8192 and (cap, un) != SWEEP[0]:
            break
    lib.pmf_debug_col(512, 4)
    if npix <= 8192:
        t_s = timeit(f_s)
        line += " | ONE LAUNCH (bn_bwd_small) %6.1f us" % t_s
    print(line, flush=True)
