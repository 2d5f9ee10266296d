#!/usr/bin/env python3
"""Host-side cost of one training iteration: wall time of the two plan launches (C loops issuing ~1200 kernels) and of
the whole step as seen by the host, with and without waiting for the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pmf_amd import plan as PL
from pmf_amd.engine import TrainEngine
from pmf_amd.models import PMFNet

acc = {}
orig = PL.Plan.run
def timed(self, ops, n, what, begin=0, end=None, sig=None):
    t = time.perf_counter(); r = orig(self, ops, n, what, begin, end, sig); acc[what] = acc.get(what, 0.0) + time.perf_counter() - t; return r
PL.Plan.run = timed
dev = torch.device("cuda", 0)
model = PMFNet(5, 3, 20, 32, imagenet_pretrained=False, image_backbone="resnet34").to(dev)
eng = TrainEngine(model, 20, lr=1e-3, warmup_steps=1000, max_steps=4900, feature_mean=bench.KITTI_MEAN, feature_std=bench.KITTI_STD)
feat0, mask, label = bench.make_batch(2, 64, 2048, 1, dev)
for _ in range(3): eng.train_step(feat0.clone(), mask, label)
torch.cuda.synchronize(); acc.clear()
K = 10
t0 = time.perf_counter()
for _ in range(K): eng.train_step(feat0.clone(), mask, label)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
for pl in model._plans.values():
    print("graphs cached: %d, keys seen: %d" % (len(pl._graphs), len(pl._graph_seen)))
print("host enqueue per step %.2f ms; GPU-complete per step %.2f ms; plan.run forward %.2f ms backward %.2f ms" % (
    (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, acc.get("forward", 0) / K * 1e3, acc.get("backward", 0) / K * 1e3))

# ---- per-phase GPU time of one iteration (events on the compute stream, host far ahead) ----
import types
ph = {}
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def train_step_timed(self, input_feature, input_mask, input_label):
    e = [ev()]
    self.model.train()
    pcd, rgb = self.prepare(input_feature, input_mask)
    label = input_label.long()
    e.append(ev())
    lidar_pred, camera_pred = self.model(pcd, rgb)
    e.append(ev())
    from pmf_amd.loss import pmf_total_loss_fused
    total, terms = pmf_total_loss_fused(lidar_pred, camera_pred, label, self.focal.alpha, self.lambda_, self.gamma, self.tau,
                                        self.focal.gamma, self.metrics.conf_matrix, self.metrics_img.conf_matrix)
    e.append(ev())
    total.backward()
    e.append(ev())
    self.optimizer.step(); self.aux_optimizer.step(); self.scheduler.step(); self.aux_scheduler.step()
    e.append(ev())
    return e
names = ["prepare", "forward", "loss", "backward", "optimizer"]
evs = []
for _ in range(8): evs.append(train_step_timed(eng, feat0.clone(), mask, label))
torch.cuda.synchronize()
for i, nme in enumerate(names):
    print("%-10s %.2f ms" % (nme, sum(e[i].elapsed_time(e[i + 1]) for e in evs[2:]) / len(evs[2:])))
print("iteration  %.2f ms (event to event)" % (sum(evs[k][0].elapsed_time(evs[k + 1][0]) for k in range(2, 7)) / 5))
