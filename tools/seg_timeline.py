#!/usr/bin/env python3
"""GPU time of the backward plan run as ONE range vs in k segments (the data-parallel path runs it in segments so that
finished gradient ranges can be all-reduced while the rest computes).  usage: seg_timeline.py [k=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_batch, KITTI_MEAN, KITTI_STD
from pmf_amd.engine import TrainEngine
from pmf_amd.models import PMFNet

def main(k):
    os.environ["PMF_DP_SEGMENTS"] = str(k)
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = PMFNet(5, 3, 20, 32, imagenet_pretrained=False, image_backbone="resnet34").to(dev)
    eng = TrainEngine(model, 20, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                      feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=1000, max_steps=4900)
    feat0, mask, label = make_batch(2, 64, 2048, 1, dev, 20)
    evs = []
    def hook(plan, op_end):
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((op_end, e))
    for mode in ("one range", "%d segments" % k):
        model._bwd_segment_hook = hook if mode != "one range" else None
        for _ in range(8):
            eng.train_step(feat0.clone(), mask, label)
        torch.cuda.synchronize()
        res = []
        t0 = time.perf_counter()
        for _ in range(20):
            evs.clear()
            s = torch.cuda.Event(enable_timing=True)
            pcd, rgb = eng.prepare(feat0.clone(), mask)
            total = eng.forward_loss(pcd, rgb, label.long())[0]
            s.record()
            total.backward()
            e = torch.cuda.Event(enable_timing=True); e.record()
            eng.optimizer.step(); eng.aux_optimizer.step()
            res.append((s, list(evs), e))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        tot = np.median([s.elapsed_time(e) for s, _, e in res[3:]])
        line = "%-12s step %.2f ms, backward %.3f ms" % (mode, wall, tot)
        if res[5][1]:
            segs = []
            for s, ev, e in res[3:]:
                prev = s; row = []
                for _, x in ev:
                    row.append(prev.elapsed_time(x)); prev = x
                segs.append(row)
            line += "  segments " + " ".join("%.3f" % v for v in np.median(np.array(segs), 0))
            line += "  cuts " + str([c for c, _ in res[5][1]])
        print(line)

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
