#!/usr/bin/env python3
"""Where the multi-lane training step could be, and what it waits for: every op of the forward and backward plans of the
bench's default configuration is timed ALONE (Plan.run_profiled: HIP events per launch, lanes off), then the plan's DAG --
lane of every op, plan events it waits for / records (pmf_op_t.pad_ bits, include/pmf_amd.h) -- is played with those
durations and NO contention between lanes: per-lane busy time, makespan, and the critical path (walked back from the op
that finishes last through whichever constraint bound every op: its lane predecessor or the event it waited for) summed
by op family.  The distance between that makespan and the measured step is what co-running kernels cost each other; the
critical-path table says which launches a change has to shorten to move the step at all (a faster side-lane kernel does
not: DESIGN.md "Round 4").

usage: python tools/critical_path.py [--height 64 --width 2048 --bs 2] [--scale family=factor ...]
  --scale conv_wgrad=0.5   what-if: play the DAG again with that family's durations multiplied by the factor"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def play(ops, lanes_on=True):
    """ops: [(lane, wait_e, rec_e, ms, family)] in list order -> (makespan, per-lane busy, critical path [(idx, why)])"""
    clock = {0: 0.0}
    ev_t, ev_src = {}, {}
    start, end, bound = [0.0] * len(ops), [0.0] * len(ops), [None] * len(ops)
    last_on_lane = {}
    busy = collections.Counter()
    for k, (lane, we, re_, ms, fam) in enumerate(ops):
        if not lanes_on:
            lane = 0
        if lane not in clock:                 # fork: the lane starts at the main lane's current position
            clock[lane] = clock[0]
            last_on_lane[lane] = last_on_lane.get(0)
        t0, why = clock[lane], ("lane", last_on_lane.get(lane))
        if lanes_on and we >= 0 and we in ev_t and ev_t[we] > t0:
            t0, why = ev_t[we], ("event", ev_src[we])
        start[k], end[k], bound[k] = t0, t0 + ms, why
        clock[lane] = end[k]
        last_on_lane[lane] = k
        busy[lane] += ms
        if re_ >= 0:
            ev_t[re_], ev_src[re_] = end[k], k
    last = max(range(len(ops)), key=lambda k: end[k])
    path, k = [], last
    while k is not None:
        path.append(k)
        k = bound[k][1]
    return max(end), dict(busy), path[::-1], start, end


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--scale", action="append", default=[])
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    from pmf_amd.models import PMFNet
    from pmf_amd.engine import TrainEngine
    from pmf_amd.utils.detinit import deterministic_init, synthetic_batch
    import numpy as np
    dev = torch.device("cuda")
    model = deterministic_init(PMFNet(5, 3, 20, 32, False, "resnet34")).to(dev).train()
    eng = TrainEngine(model, 20, alpha=np.linspace(0.2, 1.0, 20), warmup_steps=10, max_steps=1000)
    pcd, rgb, label, _ = synthetic_batch(args.bs, args.height, args.width, 20, seed=1, fill=0.15)
    pcd, rgb, label = pcd.to(dev), rgb.to(dev), label.to(dev).long()
    for _ in range(3):
        total = eng.forward_loss(pcd, rgb, label)[0]
        total.backward()
    torch.cuda.synchronize()
    plan = next(p for k, p in model._plans.items() if k[3])
    total = eng.forward_loss(pcd, rgb, label)[0]
    res = {}
    for what, arr, n in (("forward", plan.fwd_ops, plan.n_fwd), ("backward", plan.bwd_ops, plan.n_bwd)):
        if what == "backward":
            total.backward()
        plan.run_profiled(what)
        prof = plan.run_profiled(what)
        ops = []
        for k in range(n):
            bits = arr[k].pad_
            kind, fam, flops, ms, name, _ = prof[k]
            ops.append((bits & 3, ((bits >> 8) & 0xff) - 1, ((bits >> 16) & 0xff) - 1, ms, fam or kind, name))
        res[what] = ops
    scales = dict((s.split("=")[0], float(s.split("=")[1])) for s in args.scale)
    for what, ops in res.items():
        for label_, sc in (("measured", {}), ("what-if %s" % scales, scales)):
            if label_ != "measured" and not scales:
                continue
            o = [(l, w, r, ms * sc.get(fam, 1.0), fam) for (l, w, r, ms, fam, _) in ops]
            span, busy, path, start, end = play(o)
            serial = sum(x[3] for x in o)
            print("== %s (%s): %d ops, sum of alone durations %.2f ms, no-contention makespan %.2f ms; lane busy %s" % (
                what, label_, len(o), serial, span, {k: round(v, 2) for k, v in sorted(busy.items())}))
            by = collections.Counter()
            cnt = collections.Counter()
            for k in path:
                by[o[k][4]] += o[k][3]
                cnt[o[k][4]] += 1
            print("   critical path: %d ops, %.2f ms:" % (len(path), sum(by.values())),
                  ", ".join("%s %.2f (%d)" % (f, v, cnt[f]) for f, v in by.most_common(14)))
            lanes_on_path = collections.Counter(o[k][0] for k in path)
            print("   lanes of the path's ops:", dict(lanes_on_path))
            if label_ == "measured":
                top = sorted(path, key=lambda k: -o[k][3])[:args.top]
                for k in top:
                    print("      %7.1f us  lane %d  %-14s %s" % (o[k][3] * 1e3, o[k][0], o[k][4], ops[k][5][:70]))


if __name__ == "__main__":
    main()
