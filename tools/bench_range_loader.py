#!/usr/bin/env python3
"""Range-image loader throughput: HIP kernels (sweep resident on the GPU) next to the numpy oracle on one host core.
usage: bench_range_loader.py [points=120000] [H=64] [W=2048]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import range_projection_ref as RR
from oracle.cases import lidar_sweep
from pmf_amd.dataset.preprocess.projection import RangeProjection

P = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
pts, sem, lut = lidar_sweep(0, P)
rp = RangeProjection(3., -25., W, H)
dev = rp.to_device(pts)
lab = torch.as_tensor(lut[sem]).cuda()
mean = torch.tensor([12.12, 10.88, 0.23, -1.04, 0.21]).cuda()
stds = torch.tensor([12.32, 11.47, 6.91, 0.86, 0.16]).cuda()
for _ in range(20):
    rp.loader_item(dev, lab, mean, stds)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    rp.loader_item(dev, lab, mean, stds)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
# algorithmic bytes: read the sweep (16 B/point) + label (4), atomics on 8 B keys, gather reads keys + winners, writes 8 planes
alg = P * 20 + H * W * (8 + 8 + 16 + 4 + 8 * 4)
t0 = time.perf_counter()
for _ in range(3):
    RR.loader_item(pts, lut[sem], RR.fov_constants(3., -25.), H, W, [12.12, 10.88, 0.23, -1.04, 0.21], [12.32, 11.47, 6.91, 0.86, 0.16])
cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
print('{"workload": "range loader %d points -> 5x%dx%d", "gpu_us_per_sweep": %.1f, "sweeps_per_s": %.0f, '
      '"algorithmic_GBps": %.1f, "cpu_oracle_ms_per_sweep": %.1f, "cpu_cores": 1}' % (P, H, W, us, 1e6 / us, alg / us / 1e3, cpu_ms))
