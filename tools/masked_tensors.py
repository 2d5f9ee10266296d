#!/usr/bin/env python3
"""Where does a parameter-gradient difference ENTER the backward pass?  Fresh-state PMF-ResNet34 at 2 x 64 x 2048 (the batch,
Dropout2d masks and objective of tests/test_gpu_fullsize.py), tile configurations per PMF_AUTOTUNE; the float64 and fp32 oracle
passes replay the HIP path's activation decisions (oracle/act_masks.py) and upstream gradient, so all three backward passes
differentiate one piecewise-linear function.  For every conv of the plan: dz (gradient w.r.t. the conv output, Plan T.g) and
gy (gradient w.r.t. the BatchNorm output, V.gy) of the HIP path and of the fp32 oracle against float64, in backward order; for
the first tensors whose ratio jumps, where the differing elements sit.

  PMF_AUTOTUNE=0 python tools/masked_tensors.py [--kind pmf_r34] [--focus enc.layer2.2]"""
import argparse
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="pmf_r34")
    ap.add_argument("--focus", default="")
    ap.add_argument("--steps", type=int, default=0, help="training iterations on the batch before the compared pass")
    args = ap.parse_args()
    from tests import test_gpu_fullsize as TF
    from oracle import pmf_torch as O
    from oracle.act_masks import ActSites
    from pmf_amd.engine import TrainEngine
    mk_hip, mk_ref, ncls, (n, h, w), fill = TF._build(args.kind)
    from pmf_amd.utils.detinit import synthetic_batch
    ref0 = mk_ref().train()
    g = torch.Generator().manual_seed(3)
    masks = {nm: (torch.rand(n, c, generator=g) > 0.2).float() / 0.8 for nm, _, c in O.dropout_sites(ref0)}
    pcd, rgb, label, _ = synthetic_batch(n, h, w, ncls, seed=21, fill=fill)
    alpha = torch.linspace(0.2, 1.0, ncls)
    alpha[0] = 0
    hip = mk_hip().cuda().train()
    eng = TrainEngine(hip, ncls, alpha=alpha.numpy(), warmup_steps=10, max_steps=100)
    if args.steps:
        feat = torch.cat([pcd, rgb], 1).cuda()
        ones = torch.ones(n, h, w, device="cuda")
        for _ in range(args.steps):
            eng.train_step(feat.clone(), ones, label.cuda())
        torch.cuda.synchronize()
        ref0.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()})
    hip.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
    for _ in range(3):
        total, _, lp, cp, _ = eng.forward_loss(pcd.cuda(), rgb.cuda(), label.cuda().long())
        lp.retain_grad()
        cp.retain_grad()
        total.backward()
    torch.cuda.synchronize()
    up = (lp.grad.detach().cpu(), cp.grad.detach().cpu())
    plan = next(p for k, p in hip._plans.items() if k[3])
    dec = plan.act_decisions(hip)
    names = {id(m): q for q, m in hip.named_modules()}
    site_of = {names[id(conv)]: nm for conv, nm, _, _ in plan.act_sites if id(conv) in names}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    grads = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = copy.deepcopy(ref0).to(dt).train()
        O.set_dropout_masks(m, {k: v.to(dt) for k, v in masks.items()})
        cap = {}
        hs = []
        for q, mod in m.named_modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.BatchNorm2d)):
                def hook(mod_, inp, out, q=q):
                    out.register_hook(lambda gr, q=q: cap.__setitem__(q, gr.detach().double()))
                hs.append(mod.register_forward_hook(hook))
        with ActSites(m, inject=dec):
            a, b = m(pcd.to(dt), rgb.to(dt))
        torch.autograd.backward([a, b], [up[0].to(dt), up[1].to(dt)])
        grads[tag] = cap
        for hh in hs:
            hh.remove()
    # BatchNorm module behind a conv: same prefix, the reference's naming (convK -> bnK / Sequential index + 1 or + 2)
    print("%-34s %-7s %11s %11s %7s" % ("site (backward order)", "tensor", "hip vs f64", "f32 vs f64", "ratio"))
    first = []
    for conv, nm, act, relu_view in reversed(plan.act_sites):
        q = names.get(id(conv))
        if q is None or q not in grads["f64"]:
            continue
        t, v = plan.tensors[nm], plan.views[nm]
        for what, buf, ref_key in (("gy", getattr(v, "gy", None), getattr(v.bn["module"], "_q", None) if v.bn else None),
                                   ("dz", t.g, q)):
            if what == "gy":
                if v.bn is None:
                    continue
                bq = names.get(id(v.bn["module"]))
                ref_key = bq
            if buf is None or ref_key not in grads["f64"]:
                continue
            got = plan.read(buf).cpu().double()
            r64, r32 = grads["f64"][ref_key], grads["f32"][ref_key]
            if got.shape != r64.shape:
                continue
            den = r64.norm().clamp_min(1e-30)
            eh, ec = float((got - r64).norm() / den), float((r32 - r64).norm() / den)
            flag = "  <--" if eh > 4 * max(ec, 1e-7) else ""
            print("%-34s %-7s %11.3e %11.3e %7.2f%s" % (nm, what, eh, ec, eh / max(ec, 1e-30), flag), flush=True)
            if (flag and len(first) < 3) or (args.focus and nm.startswith(args.focus)):
                first.append((nm, what, got, r64, r32))
    for nm, what, got, r64, r32 in first:
        d = (got - r64).abs()
        thr = 1e-3 * float(r64.abs().max())
        big = d > thr
        print("\n== %s %s: shape %s, max |ref| %.3e, max |diff| %.3e, elements beyond 1e-3 of max: %d (fp32 oracle: %d)" % (
            nm, what, tuple(got.shape), float(r64.abs().max()), float(d.max()), int(big.sum()),
            int(((r32 - r64).abs() > thr).sum())))
        idx = big.nonzero()
        if idx.numel():
            for dim, lab in enumerate(("n", "c", "y", "x")):
                vals, cnt = idx[:, dim].unique(return_counts=True)
                top = sorted(zip(cnt.tolist(), vals.tolist()), reverse=True)[:12]
                print("   by %s: %d distinct; top %s" % (lab, vals.numel(), top))
        e_pix = d.square().sum(1).sqrt()       # [n, y, x]
        flat = e_pix.flatten()
        topv, topi = flat.topk(8)
        H, W = e_pix.shape[1], e_pix.shape[2]
        print("   worst pixels (n, y, x, |diff|_2, |ref|_2):",
              [(int(i // (H * W)), int(i % (H * W) // W), int(i % W), float(vv),
                float(r64[int(i // (H * W)), :, int(i % (H * W) // W), int(i % W)].norm())) for vv, i in zip(topv, topi)])
        print("   share of the squared error in the worst 0.1 %% of pixels: %.3f" % float(
            flat.square().topk(max(1, flat.numel() // 1000))[0].sum() / flat.square().sum()))


if __name__ == "__main__":
    main()
