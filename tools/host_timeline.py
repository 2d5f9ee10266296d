#!/usr/bin/env python3
"""Where one training iteration spends its time on the GPU timeline vs on the host: HIP events between the phases of
TrainEngine.train_step (prepare / forward plan / objective / backward plan / optimizers) and perf_counter stamps of the
host code that enqueues them.  GPU phase time >> the phase's kernel time = the stream starved (host-bound).
usage: python tools/host_timeline.py [steps=20]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import make_batch, KITTI_MEAN, KITTI_STD
from pmf_amd.engine import TrainEngine
from pmf_amd.models import PMFNet


def main(steps):
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = PMFNet(5, 3, 20, 32, imagenet_pretrained=False, image_backbone="resnet34").to(dev)
    eng = TrainEngine(model, 20, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                      feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=1000, max_steps=4900)
    feat0, mask, label = make_batch(2, 64, 2048, 1, dev, 20)
    for _ in range(8):
        eng.train_step(feat0.clone(), mask, label)
    torch.cuda.synchronize()
    names = ["prepare", "forward", "objective", "backward", "optimizers", "rest"]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(steps)]
    host = np.zeros((steps, len(names)))
    t_all0 = time.perf_counter()
    for s in range(steps):
        e = ev[s]
        t = [time.perf_counter()]
        e[0].record()
        eng.model.train()
        pcd, rgb = eng.prepare(feat0.clone(), mask)
        lab = label.long()
        e[1].record(); t.append(time.perf_counter())
        lidar_pred, camera_pred = eng.model(pcd, rgb)
        e[2].record(); t.append(time.perf_counter())
        from pmf_amd.loss import pmf_total_loss_fused
        total, terms = pmf_total_loss_fused(lidar_pred, camera_pred, lab, eng.focal.alpha, eng.lambda_, eng.gamma, eng.tau,
                                            eng.focal.gamma, eng.metrics.conf_matrix, eng.metrics_img.conf_matrix)
        eng.metrics.external_update()
        eng.metrics_img.external_update()
        e[3].record(); t.append(time.perf_counter())
        total.backward()
        e[4].record(); t.append(time.perf_counter())
        eng.optimizer.step(); eng.aux_optimizer.step(); eng.scheduler.step(); eng.aux_scheduler.step()
        e[5].record(); t.append(time.perf_counter())
        eng.iteration += 1
        e[6].record(); t.append(time.perf_counter())
        host[s] = np.diff(t)
    t_enq = time.perf_counter() - t_all0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all0
    gpu = np.array([[ev[s][i].elapsed_time(ev[s][i + 1]) for i in range(len(names))] for s in range(steps)])
    step_gap = np.array([ev[s][6].elapsed_time(ev[s + 1][0]) for s in range(steps - 1)])
    print("steps %d: wall %.2f ms/step, host enqueue %.2f ms/step" % (steps, t_all / steps * 1e3, t_enq / steps * 1e3))
    print("%-12s %10s %10s" % ("phase", "gpu ms", "host ms"))
    for i, n in enumerate(names):
        print("%-12s %10.3f %10.3f" % (n, np.median(gpu[3:, i]), 1e3 * np.median(host[3:, i])))
    print("between steps (gpu): %.3f ms" % np.median(step_gap[3:]))
    print("first 3 steps host ms (GPU idle at start):", (1e3 * host[:3].sum(1)).round(2))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
