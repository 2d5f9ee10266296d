#!/usr/bin/env python3
"""Where does a long-trained state lose accuracy?  Trains the bench's headline configuration for --steps iterations on one
batch (what `bench.py --steps N` does), copies the state into the CPU oracle and compares, tensor by tensor in the oracle's
execution order, the HIP forward pass and the fp32 CPU oracle's forward pass against the float64 oracle (relative L2 per tensor):
the first tensor whose HIP / fp32-oracle ratio jumps names the layer.  Round 5: the 1500-iteration parity block failed with the
wave-scheduled conv kernel in the plan (every LiDAR-stream gradient 3-15x the fp32 oracle's distance from float64).

  python tools/soak_tensors.py --steps 1500 [--ws 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--backbone", default="resnet34")
    ap.add_argument("--nclasses", type=int, default=20)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--masked", action="store_true", help="both oracle passes replay the HIP path's activation decisions")
    args = ap.parse_args()
    import bench as B
    from oracle import pmf_torch as O
    from pmf_amd.engine import TrainEngine
    from pmf_amd.models import PMFNet
    from pmf_amd.utils.detinit import deterministic_init
    from tests import gpu_helpers as G
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    torch.cuda.manual_seed(1)
    model = deterministic_init(PMFNet(5, 3, args.nclasses, 32, imagenet_pretrained=False, image_backbone=args.backbone)).to(dev)
    eng = TrainEngine(model, args.nclasses, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                      feature_mean=B.KITTI_MEAN, feature_std=B.KITTI_STD, warmup_steps=10 * 100, max_steps=49 * 100)
    feat0, mask, label = B.make_batch(2, args.height, args.width, 1, dev, args.nclasses)
    for _ in range(args.steps):
        eng.train_step(feat0.clone(), mask, label)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    masks = {n: (torch.rand(2, c, generator=g) > 0.2).float() / 0.8 for n, c in model._mask_sites()}
    model.set_dropout_masks({k: v.to(dev) for k, v in masks.items()})
    model.train()
    pcd, rgb = eng.prepare(feat0.clone(), mask)
    total, _, lp, cp, _ = eng.forward_loss(pcd, rgb, label.long())
    lp.retain_grad(); cp.retain_grad()
    total.backward()
    torch.cuda.synchronize()
    gl_h, gc_h = lp.grad.detach().cpu().double(), cp.grad.detach().cpu().double()
    plan = next(p for k, p in model._plans.items() if k[3])
    dec = plan.act_decisions(model) if args.masked else None
    caps = {}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        net = O.PMFNet(5, 3, args.nclasses, 32, False, args.backbone)
        net.load_state_dict(sd)
        net = net.to(dt).train()
        O.set_dropout_masks(net, {k: v.to(dt) for k, v in masks.items()})
        cap, hs = G.capture_oracle(net)
        e = TrainEngine(net, args.nclasses, lambda_=1.0, gamma=0.5, tau=0.7, feature_mean=B.KITTI_MEAN, feature_std=B.KITTI_STD,
                        warmup_steps=10, max_steps=100)
        e.focal.to(dt)
        p, r = e.prepare(feat0.detach().cpu().to(dt), mask.cpu().to(dt))
        if args.masked:
            from oracle.act_masks import ActSites
            with ActSites(net, inject=dec):
                tot, _, lpo, cpo, _ = e.forward_loss(p, r, label.cpu().long())
        else:
            tot, _, lpo, cpo, _ = e.forward_loss(p, r, label.cpu().long())
        own = torch.autograd.grad(tot, [lpo, cpo], retain_graph=True)
        for v in cap.values():
            if isinstance(v, torch.Tensor) and v.requires_grad and not v.is_leaf:
                v.retain_grad()
        # the network's backward pass driven by the HIP path's objective gradient (the objective's own kinks stay out of it)
        torch.autograd.backward([lpo, cpo], [gl_h.to(dt), gc_h.to(dt)])
        caps[tag] = {k: v.detach().double() for k, v in cap.items()}
        caps[tag + ".grad"] = {k: v.grad.detach().double() for k, v in cap.items() if isinstance(v, torch.Tensor) and v.grad is not None}
        caps[tag + ".pgrad"] = {k: q.grad.detach().double() for k, q in net.named_parameters() if q.grad is not None}
        caps[tag + ".g"] = (own[0].detach().double(), own[1].detach().double(), float(tot))
        for h in hs:
            h.remove()
    # the gradient of the objective w.r.t. the two probability maps: the objective is discontinuous (confidence thresholds of the
    # perception-aware terms, the Lovasz ranking) -- a rounding-level difference in the probabilities can move it by whole terms
    g64l, g64c, l64 = caps.pop("f64.g")
    g32l, g32c, l32 = caps.pop("f32.g")
    for nm, gh, g32, g64 in (("d objective / d lidar probabilities", gl_h, g32l, g64l), ("d objective / d camera probabilities", gc_h, g32c, g64c)):
        den = g64.norm().clamp_min(1e-30)
        big = lambda a: int(((a - g64).abs() > 1e-3 * g64.abs().max()).sum())
        print("%-38s hip vs f64 %.3e (%d elements beyond 1e-3 of the maximum)   f32 vs f64 %.3e (%d)" % (
            nm, float((gh - g64).norm() / den), big(gh), float((g32 - g64).norm() / den), big(g32)), flush=True)
    print("loss: hip %.7f  f32 %.7f  f64 %.7f" % (float(total), l32, l64))
    print("%-22s %12s %12s %8s" % ("tensor", "hip vs f64", "f32 vs f64", "ratio"))
    print("---- gradients w.r.t. the same tensors (all three backward passes driven by the HIP path's objective gradient)")
    for name, ref in caps["f64.grad"].items():
        t = None
        if name in plan.views and getattr(plan.views[name], "gy", None) is not None:
            t = plan.views[name].gy
        elif name in plan.tensors and getattr(plan.tensors[name], "g", None) is not None:
            t = plan.tensors[name].g
        if t is None or name not in caps["f32.grad"]:
            continue
        got = plan.read(t).cpu().double()
        if got.shape != ref.shape:
            continue
        den = ref.norm().clamp_min(1e-30)
        eh, ec = float((got - ref).norm() / den), float((caps["f32.grad"][name] - ref).norm() / den)
        print("grad %-17s %12.3e %12.3e %8.2f" % (name, eh, ec, eh / max(ec, 1e-30)), flush=True)
    print("---- parameter gradients, worst ratios")
    rows = []
    for k, q in model.named_parameters():
        if q.grad is None or k not in caps["f64.pgrad"]:
            continue
        ref = caps["f64.pgrad"][k]
        den = ref.norm().clamp_min(1e-30)
        eh, ec = float((q.grad.detach().cpu().double() - ref).norm() / den), float((caps["f32.pgrad"][k] - ref).norm() / den)
        rows.append((eh / max(ec, 1e-30), k, eh, ec))
    for r in sorted(rows, reverse=True)[:25]:
        print("param %-55s %10.3e %10.3e %8.2f" % (r[1], r[2], r[3], r[0]), flush=True)
    print("---- forward tensors")
    for name, ref in caps["f64"].items():
        if name in plan.views:
            got = plan.read_view(plan.views[name]).cpu().double()
        elif name in plan.tensors:
            got = plan.read(plan.tensors[name]).cpu().double()
        else:
            continue
        if got.shape != ref.shape:
            continue
        den = ref.norm().clamp_min(1e-30)
        eh = float((got - ref).norm() / den)
        ec = float((caps["f32"][name] - ref).norm() / den)
        print("%-22s %12.3e %12.3e %8.2f" % (name, eh, ec, eh / max(ec, 1e-30)), flush=True)


if __name__ == "__main__":
    main()
