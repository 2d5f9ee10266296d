#!/usr/bin/env python3
"""The fused objective alone (loss/fused.py: loss_pixel_k + in-library Lovasz sort + gradient scatter, both heads, forward and
backward) on the bench's label map -- for rocprofv3 --kernel-trace --stats of the loss kernels.
usage: python tools/bench_loss.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pmf_amd.loss import pmf_total_loss_fused
from pmf_amd.utils.detinit import synthetic_batch, det_tensor
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n, c, h, w = 2, 20, 64, 2048
_, _, label, _ = synthetic_batch(n, h, w, c, seed=1)
label = label.cuda()
alpha = np.ones(c, np.float32); alpha[0] = 0
a = torch.softmax(det_tensor("bl.a", (n, c, h, w), -3, 3), 1).cuda().requires_grad_(True)
b = torch.softmax(det_tensor("bl.b", (n, c, h, w), -3, 3), 1).cuda().requires_grad_(True)
al = torch.from_numpy(alpha)
def step():
    a.grad = b.grad = None
    tot, _ = pmf_total_loss_fused(a, b, label, al)
    tot.backward()
    return tot
for _ in range(10): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): t = step()
e1.record(); torch.cuda.synchronize()
print("fused objective fwd+bwd: %.1f us per call, labelled fraction %.3f, loss %.6f" % (
    e0.elapsed_time(e1) / reps * 1e3, (label > 0).float().mean().item(), t.item()))
# exact fingerprints (the kernels are deterministic: equal across runs and across PMF_LOSS_XCD_ROWS=0 / 1)
print("fingerprint loss %r grad_a %r %r grad_b %r %r" % (
    t.item(), a.grad.double().sum().item(), a.grad.double().abs().sum().item(),
    b.grad.double().sum().item(), b.grad.double().abs().sum().item()))
