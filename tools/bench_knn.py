#!/usr/bin/env python3
"""KNN vote kernel alone, timed on the GPU (a torch CUDA graph of REPS launches: the ctypes launch path costs ~20 us of host
time per call, more than the kernel in sweep order).  usage: python tools/bench_knn.py [sweep|random]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmf_amd.postproc import KNN
order = sys.argv[1] if len(sys.argv) > 1 else "sweep"
B, H, W, REPS = 4, 64, 2048, 50
g = torch.Generator().manual_seed(3)
dev = torch.device("cuda")
knn = KNN({"knn": 5, "search": 5, "sigma": 1.0, "cutoff": 1.0}, 20)
pr = torch.rand(B, H, W, generator=g) * 50 + 2
mask = torch.rand(B, H, W, generator=g) < 0.15
pr = torch.where(mask, pr, torch.full_like(pr, -1.0))
am = torch.randint(0, 20, (B, H, W), generator=g)
frames = []
for b in range(B):
    occ = torch.nonzero(mask[b])                       # row-major (y, x)
    sel = torch.randint(0, occ.shape[0], (int(occ.shape[0] * 1.3),), generator=g)
    if order == "sweep":                               # azimuth by azimuth: column-major
        key = occ[sel, 1] * H + occ[sel, 0]
        sel = sel[torch.argsort(key, stable=True)]
    py, px = occ[sel, 0], occ[sel, 1]
    ur = pr[b, py, px] + torch.randn(sel.shape[0], generator=g) * 0.2
    frames.append((ur, px, py))
n = [f[0].shape[0] for f in frames]
off = torch.tensor([0] + list(torch.tensor(n).cumsum(0)), dtype=torch.int64, device=dev)
args = (pr.to(dev), am.to(dev), torch.cat([f[0] for f in frames]).to(dev), torch.cat([f[1] for f in frames]).to(dev),
        torch.cat([f[2] for f in frames]).to(dev), off)
fn, out = knn.bind_batch(*args)
for _ in range(20): fn()
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    fn2, out2 = knn.bind_batch(*args)
    fn2(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(REPS): fn2()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): gr.replay()
    e1.record(s); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / (10 * REPS) * 1e3
P = sum(n)
mb = (12.0 * H * W * B + 28.0 * P) / 1e6
print("knn %s order: %d points in %d frames, %.2f us per launch (graph of %d), %.2f MB algorithmic -> %.0f GB/s = %.3f of 8 TB/s; labels checksum %d"
      % (order, P, B, us, REPS, mb, mb / us * 1e3 / 1e3 * 1e3, mb / us / 8000.0 * 1e3 / 1e3 * 1e3, int(out2.sum().item())))
