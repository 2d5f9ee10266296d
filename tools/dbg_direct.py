#!/usr/bin/env python3
"""Debug: which conv op changes the gradients when it alone runs the direct multi-tap variant (cfg bit 24)?
For every multi-tap split-bf16 conv op of the forward / backward plan (2x64x2048 by default): force the variant on that op
only, run forward + backward, compare every parameter gradient with the heuristic plan."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMF_AUTOTUNE"] = "0"
import torch
from pmf_amd import _lib as L
from pmf_amd.models import PMFNet
from pmf_amd.engine import TrainEngine
from pmf_amd.utils.detinit import deterministic_init, synthetic_batch
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 2048)
hip = deterministic_init(PMFNet(5, 3, 20, 32, False, "resnet34")).cuda().train()
eng = TrainEngine(hip, 20, warmup_steps=10, max_steps=100)
pcd, rgb, label, _ = synthetic_batch(2, h, w, 20, seed=21, fill=0.25)
g = torch.Generator().manual_seed(3)
hip.set_dropout_masks({n: ((torch.rand(2, c, generator=g) > 0.2).float() / 0.8).cuda() for n, c in hip._mask_sites()})
d_pcd, d_rgb, d_label = pcd.cuda(), rgb.cuda(), label.cuda().long()
os.environ["PMF_GRAPH"] = "0"
def grads():
    total = eng.forward_loss(d_pcd, d_rgb, d_label)[0]
    total.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in hip.named_parameters()}, float(total)
base, l0 = grads()
plan = next(p for k, p in hip._plans.items() if k[3])
lib = L.lib()
cands = []
for tag, ops, n, kinds, shift, meta, fins in (("fwd", plan.fwd_ops, plan.n_fwd, plan.fwd_kinds, plan.fwd_shift, plan.meta_fwd, plan._conv_fin),
                                        ("bwd", plan.bwd_ops, plan.n_bwd, plan.bwd_kinds, plan.bwd_shift, plan.meta_bwd, plan._conv_fold)):
    for k in range(n):
        if kinds[k] == L.OP_CONV and ops[k].u.conv.w_s3 and 1 < ops[k].u.conv.ntaps <= 9:
            cands.append((tag, ops, k, meta.get(k - shift, {}), fins.get(k - shift), shift))
print("candidates", len(cands), "loss", l0, flush=True)
for tag, ops, k, m, fin, shift in cands:
    d = ops[k].u.conv
    for cfg in (32 | (1 << 8) | (1 << 16) | (1 << 24), 64 | (2 << 8) | (1 << 16) | (1 << 24)):
        if (cfg & 255) == 64 and d.Cout <= 32: continue
        d.cfg = cfg
        flist = fin if isinstance(fin, list) else ([] if fin is None else [fin])
        for fi in flist: ops[fi + shift].u.sm.i[1] = lib.pmf_conv_fwd_stat_rows(C.byref(d))
        got, l1 = grads()
        def rel(n):
            wk = n.rsplit(".", 1)[0] + ".weight"
            floor = 1e-3 * base[wk].norm().item() if wk in base else 0.0
            return ((got[n] - base[n]).norm() / max(base[n].norm().item(), floor, 1e-20)).item()
        wn = max(base, key=rel)
        worst = rel(wn)
        flag = "  <<<<<<" if worst > 2e-4 else ""
        print("%s #%d %-18s %-34s cfg %#x nsrc %d  worst dgrad %.2e (%s) dloss %.3e%s" % (
            tag, k, m.get("name"), m.get("shape"), cfg, d.nsrc, worst, wn, abs(l1 - l0), flag), flush=True)
    d.cfg = 0
    for fi in flist: ops[fi + shift].u.sm.i[1] = lib.pmf_conv_fwd_stat_rows(C.byref(d))
