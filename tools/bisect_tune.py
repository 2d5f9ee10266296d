"""Which tuned tile configuration moves a plan's gradients?  (round 5: the autotuned EPMF-R34 plan put the low-resolution
layers of the camera decoder 1e-3 away from float64 where the heuristic plan and the fp32 CPU oracle sit at 1e-5.)

Builds the training plan of one full-size configuration with the tuner's choices, takes the HEURISTIC plan's gradients as
the yardstick (they pass the float64 bars of tests/test_gpu_fullsize.py) and bisects over the conv launches whose tuned
configuration differs from 0: the smallest set of launches that moves the probe gradients by more than --bar.

  python tools/bisect_tune.py --kind epmf [--probe camera_stream_decoder.up_2a.0.bias]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmf_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="epmf")
    ap.add_argument("--probe", default="camera_stream_decoder.up_2a.0.bias,camera_stream_decoder.aspp.conv.weight")
    ap.add_argument("--bar", type=float, default=1e-4)
    args = ap.parse_args()
    from tests.test_gpu_fullsize import _build
    from pmf_amd.engine import TrainEngine
    from pmf_amd.utils.detinit import synthetic_batch
    from oracle import pmf_torch as O
    mk_hip, mk_ref, ncls, (n, h, w), fill = _build(args.kind)
    ref = mk_ref()
    g = torch.Generator().manual_seed(3)
    masks = {nm: (torch.rand(n, c, generator=g) > 0.2).float() / 0.8 for nm, _, c in O.dropout_sites(ref)}
    del ref
    pcd, rgb, label, _ = synthetic_batch(n, h, w, ncls, seed=21, fill=fill)
    alpha = torch.linspace(0.2, 1.0, ncls)
    alpha[0] = 0
    hip = mk_hip().cuda().train()
    eng = TrainEngine(hip, ncls, alpha=alpha.numpy(), warmup_steps=10, max_steps=100)
    hip.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
    d = (pcd.cuda(), rgb.cuda(), label.cuda().long())
    probes = args.probe.split(",")
    params = dict(hip.named_parameters())

    def grads():
        for _ in range(2):
            total = eng.forward_loss(*d)[0]
            total.backward()
        torch.cuda.synchronize()
        return {k: params[k].grad.detach().clone() for k in probes}

    grads()
    plan = next(p for k, p in hip._plans.items() if k[3])
    lib = L.lib()
    sites = []      # (ops, k, shift, fins, tuned cfg, label)
    for what, ops, nn, kinds, shift, fins, meta in (
            ("fwd", plan.fwd_ops, plan.n_fwd, plan.fwd_kinds, plan.fwd_shift, plan._conv_fin, plan.meta_fwd),
            ("bwd", plan.bwd_ops, plan.n_bwd, plan.bwd_kinds, plan.bwd_shift, plan._conv_fold, plan.meta_bwd)):
        for k in range(nn):
            if kinds[k] == L.OP_CONV and ops[k].u.conv.cfg != 0:
                m = meta.get(k - shift, {})
                sites.append((ops, k, shift, fins, int(ops[k].u.conv.cfg), "%s #%d %s [%s]" % (what, k, m.get("name", ""), m.get("shape", ""))))
    print("%d conv launches run a tuned configuration" % len(sites), flush=True)

    def apply(on):
        on = set(on)
        for i, (ops, k, shift, fins, cfg, _) in enumerate(sites):
            dd = ops[k].u.conv
            dd.cfg = cfg if i in on else 0
            fin = fins.get(k - shift)
            for fi in (fin if isinstance(fin, list) else ([] if fin is None else [fin])):
                ops[fi + shift].u.sm.i[1] = lib.pmf_conv_fwd_stat_rows(C.byref(dd))
        for gr in plan._graphs.values():
            lib.pmf_graph_destroy(gr)
        plan._graphs.clear()
        plan._graph_seen.clear()

    def dist(a, b):
        return max(((a[k] - b[k]).norm() / b[k].norm().clamp_min(1e-30)).item() for k in probes)

    apply([])
    base = grads()
    again = grads()
    print("heuristic plan twice: %.2e" % dist(again, base), flush=True)
    apply(range(len(sites)))
    print("all tuned: %.2e" % dist(grads(), base), flush=True)
    cand = list(range(len(sites)))
    while len(cand) > 1:
        half = cand[:len(cand) // 2]
        apply(half)
        e = dist(grads(), base)
        print("  %3d launches [%d..%d]: %.2e" % (len(half), half[0], half[-1], e), flush=True)
        if e > args.bar:
            cand = half
        else:
            rest = cand[len(cand) // 2:]
            apply(rest)
            e2 = dist(grads(), base)
            print("  %3d launches [%d..%d]: %.2e" % (len(rest), rest[0], rest[-1], e2), flush=True)
            if e2 <= args.bar:
                print("neither half alone moves the probes: an interaction; stopping at", [sites[i][5] for i in cand][:8])
                break
            cand = rest
    for i in cand[:4]:
        print("culprit: %s cfg 0x%x" % (sites[i][5], sites[i][4]))
        apply([i])
        print("   alone: %.2e" % dist(grads(), base))
        apply([j for j in range(len(sites)) if j != i])
        print("   all but it: %.2e" % dist(grads(), base))


if __name__ == "__main__":
    main()
