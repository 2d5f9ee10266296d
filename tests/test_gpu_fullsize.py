"""-m gpu: parity of the TIMED plans at BASELINE.json's own sizes (VERDICT r03 "what's missing" 3 and 4).

* backward of configs[2] (PMF-ResNet34, 2 x 64 x 2048), configs[3] (PMF-ResNet50, 17 classes, 2 x 32 x 1024) and
  configs[4] (EPMF-ResNet34, 2 x 64 x 2048): every parameter gradient of the flat training state -- the path bench.py
  times: TrainEngine, fused objective, lanes, hipGraph replay -- against the CPU oracle in float64 with the fp32 CPU
  oracle as the yardstick, with the heuristic tile configurations and with the autotuner on
  (tasks/pmf/trainer.py:214-219 is the reference's backward);
* configs[1] at its batch size: eval logits of 4 frames + the KNN labels of every frame against the oracle
  (tasks/pmf_eval_semantickitti/infer.py:67-160).

No outlier allowance and no "worst error anywhere" yardstick: every parameter is held to the fp32 CPU
oracle's own distance from float64 for THAT parameter.
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pmf_amd.utils.detinit import deterministic_init, synthetic_batch  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402

# the two heads: no BatchNorm lies behind them in backward order
HEADS = ("lidar_stream.logits.", "camera_stream_decoder.conv.")

_ORACLE = {}


def _build(kind):
    from oracle import pmf_torch as O
    if kind == "epmf":
        from pmf_amd.models import EPMFNet
        from oracle import epmf_torch as E
        return (lambda: deterministic_init(EPMFNet(5, 3, 20, 32, False, "resnet34")),
                lambda: deterministic_init(E.EPMFNet(5, 3, 20, 32, False, "resnet34")), 20, (2, 64, 2048), 0.3)
    from pmf_amd.models import PMFNet
    if kind == "pmf_r34_sb":
        # S_B of SURVEY 8d: both streams 480 x 640 (the "+ 480x640" of BASELINE.json's metric).  H / 16 = 30 rows at the
        # bottleneck: the only BASELINE shape where all nine taps of the dilation-12 / 18 ASPP branches are in bounds
        # (pc_processor/models/pmf_net.py:110-115), forward and backward
        return (lambda: deterministic_init(PMFNet(5, 3, 20, 32, False, "resnet34")),
                lambda: deterministic_init(O.PMFNet(5, 3, 20, 32, False, "resnet34")), 20, (2, 480, 640), 0.25)
    if kind == "r50_sb":
        # BASELINE configs[3] at its RGB size: PMF-ResNet50, 17 classes, both streams 480 x 640 (pmf_net.py:50-51,85-88;
        # tasks/pmf/config_server_nus.yaml): 1024 / 2048-channel Bottleneck layers on 30 x 40 / 15 x 20 maps
        return (lambda: deterministic_init(PMFNet(5, 3, 17, 32, False, "resnet50")),
                lambda: deterministic_init(O.PMFNet(5, 3, 17, 32, False, "resnet50")), 17, (2, 480, 640), 0.25)
    if kind == "r50":
        return (lambda: deterministic_init(PMFNet(5, 3, 17, 32, False, "resnet50")),
                lambda: deterministic_init(O.PMFNet(5, 3, 17, 32, False, "resnet50")), 17, (2, 32, 1024), 0.25)
    return (lambda: deterministic_init(PMFNet(5, 3, 20, 32, False, "resnet34")),
            lambda: deterministic_init(O.PMFNet(5, 3, 20, 32, False, "resnet34")), 20, (2, 64, 2048), 0.25)


def _oracle_grads(kind):
    """fp32 and float64 CPU oracle gradients of the 5-term objective (cached: the two tile-configuration runs share them)"""
    if kind in _ORACLE:
        return _ORACLE[kind]
    from oracle import pmf_torch as O
    from oracle import losses_ref
    _, mk_ref, ncls, (n, h, w), fill = _build(kind)
    ref = mk_ref().train()
    g = torch.Generator().manual_seed(3)
    masks = {nm: (torch.rand(n, c, generator=g) > 0.2).float() / 0.8 for nm, _, c in O.dropout_sites(ref)}
    pcd, rgb, label, _ = synthetic_batch(n, h, w, ncls, seed=21, fill=fill)
    alpha = torch.linspace(0.2, 1.0, ncls)
    alpha[0] = 0
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = copy.deepcopy(ref).to(dt)
        O.set_dropout_masks(m, {k: v.to(dt) for k, v in masks.items()})
        a, b = m(pcd.to(dt), rgb.to(dt))
        tot, _ = losses_ref.pmf_total_loss(a, b, label, alpha.to(dt))
        tot.backward()
        out[tag] = ({k: p.grad.detach().clone() for k, p in m.named_parameters()}, float(tot.detach()),
                    m.lidar_stream.last_logits.detach().clone())
        del m, a, b, tot
    _ORACLE[kind] = (out, masks, (pcd, rgb, label), alpha)
    return _ORACLE[kind]


# (the headline configuration with both tile-configuration sources; configs[3] / [4] with the reproducible heuristics)
@pytest.mark.parametrize("kind,tune", [("pmf_r34", "0"), ("pmf_r34", "cache"), ("r50", "0"), ("r50", "cache"), ("epmf", "0"),
                                       ("epmf", "cache"), ("pmf_r34_sb", "0"), ("pmf_r34_sb", "cache"), ("r50_sb", "0"),
                                       ("r50_sb", "cache")],
                         ids=["pmf_r34-heuristic", "pmf_r34-shipped", "r50-heuristic", "r50-shipped", "epmf-heuristic",
                              "epmf-shipped", "pmf_r34_sb-heuristic", "pmf_r34_sb-shipped", "r50_sb-heuristic", "r50_sb-shipped"])
def test_full_size_backward_vs_oracle(kind, tune):
    from pmf_amd.engine import TrainEngine
    from pmf_amd import plan as PL
    mk_hip, _, ncls, (n, h, w), _ = _build(kind)
    (out, masks, (pcd, rgb, label), alpha) = _oracle_grads(kind)
    g64, loss64, logits64 = out["f64"]
    g32, _, _ = out["f32"]
    old = os.environ.get("PMF_AUTOTUNE")
    os.environ["PMF_AUTOTUNE"] = tune
    try:
        hip = mk_hip().cuda().train()
        eng = TrainEngine(hip, ncls, alpha=alpha.numpy(), warmup_steps=10, max_steps=100)
        assert eng.flat is not None
        hip.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
        d_pcd, d_rgb, d_label = pcd.cuda(), rgb.cuda(), label.cuda()
        # three passes: eager, eager + capture, hipGraph replay -- the gradients compared are those of the replay
        for _ in range(3):
            total = eng.forward_loss(d_pcd, d_rgb, d_label.long())[0]
            total.backward()
        torch.cuda.synchronize()
        plan = next(p for k, p in hip._plans.items() if k[3])
        assert len(plan._graphs) >= 2, "the compared pass did not run on captured graphs"
        if tune == "cache":     # the shipped table was applied: most conv launches of a BASELINE shape carry a tuned configuration
            from pmf_amd import _lib as L
            tuned = sum(1 for k in range(plan.n_fwd) if plan.fwd_kinds[k] == L.OP_CONV and plan.fwd_ops[k].u.conv.cfg != 0)
            assert tuned > 20, tuned
    finally:
        if old is None:
            os.environ.pop("PMF_AUTOTUNE", None)
        else:
            os.environ["PMF_AUTOTUNE"] = old
    assert abs(float(total) - loss64) < 1e-4 * max(1.0, abs(loss64))
    assert G.rel_err(plan.read(plan.tensors["logits"]).cpu().numpy(), logits64.float().numpy()) < 1e-3
    rows = []
    for k, p in hip.named_parameters():
        assert p.grad is not None, k
        ref = g64[k]
        wk = k.rsplit(".", 1)[0] + ".weight"
        # (a conv bias in front of a train-mode BatchNorm has a true gradient of exactly 0: measure it on the scale of
        # its layer's weight gradient instead of on its own rounding noise)
        floor = 1e-6 * g64[wk].norm().item() if wk in g64 else 0.0
        den = max(ref.norm().item(), floor, 1e-30)
        rows.append((k, (p.grad.cpu().double() - ref).norm().item() / den, (g32[k].double() - ref).norm().item() / den))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "fullsize_grads_%s_tune%s.txt" % (kind, tune)), "w") as f:
        for r in rows:
            f.write("%-60s %.3e %.3e\n" % r)
    ratio = np.array([r[1] / max(r[2], 1e-12) for r in rows if r[2] > 1e-7])
    gmean, p90 = float(np.exp(np.log(np.maximum(ratio, 1e-6)).mean())), float(np.percentile(ratio, 90))
    worst = max(rows, key=lambda r: r[1])
    print("[fullsize %s tune=%s] worst %s %.2e (cpu fp32 %.2e), ratio gmean %.2f p90 %.2f" % (
        kind, tune, worst[0], worst[1], worst[2], gmean, p90))
    # Measured at 2 x 64 x 2048 (PMF-R34, hash init): the fp32 CPU oracle itself sits 1e-6 (heads) ... 1.5e-2 (camera
    # encoder) ... 1.6e-1 (a conv bias in front of a train-mode BatchNorm, true gradient ~0) away from float64 -- the
    # BatchNorm backward subtracts two per-channel means from gy (cancellation) in every one of ~90 layers -- and the HIP
    # path tracks it parameter by parameter (ratio geometric mean 0.90, 90th percentile 1.07).  So the yardstick for a
    # parameter is the fp32 CPU oracle's own distance from float64:
    # (1) the two heads (no BatchNorm behind them in backward order): fp32 rounding level
    for k, e_h, _ in rows:
        if k.startswith(HEADS):
            assert e_h < 2e-5, (k, e_h)
    # (2) EVERY parameter: at most 3x as far from float64 as the fp32 CPU oracle for THAT parameter; no outlier allowance.
    # Floor: a fixed 2e-4 (round 5; rounds 3-4 used 5 % of the network-wide fp32 noise level, ~7.5e-4).  What the floor cannot
    # absorb is a ReLU sitting on its kink: two valid fp32 roundings of this network differ in the sign of a few
    # pre-activations per tensor, and one such flip in the camera decoder's 16 x 512 map moved these gradients by 1-4e-3 while
    # the fp32 CPU oracle sat at 1e-5 (tools/bisect_tune.py --flips).  The shipped tile table is chosen
    # under that constraint (tools/bisect_tune.py --fix), so both plans tested here are deterministic and inside the bar.
    bad = [r for r in rows if not r[1] <= max(3 * r[2], 2e-4)]
    assert not bad, "gradient error vs float64 (hip, cpu-fp32):\n" + "\n".join("%-55s %.3e %.3e" % r for r in bad[:30])
    # (3) no systematic excess over the fp32 CPU path
    assert gmean < 1.25 and p90 < 1.6, (gmean, p90)


@pytest.mark.parametrize("h,w,backbone,ncls", [(64, 2048, "resnet34", 20), (480, 640, "resnet34", 20), (512, 640, "resnet50", 17)],
                         ids=["S_A-64x2048", "S_B-480x640", "S_G-512x640-r50"])
def test_infer_bs4_logits_and_knn_vs_oracle(h, w, backbone, ncls):
    """BASELINE configs[1]: eval forward of FOUR frames in one call + the KNN vote of every frame, at both shapes the
    metric names (S_A: both streams 64 x 2048; S_B: both 480 x 640) and for configs[3]'s network (PMF-ResNet50,
    17 classes) at the nuScenes evaluation size 512 x 640"""
    from pmf_amd.models import PMFNet
    from pmf_amd.postproc import KNN
    from oracle import pmf_torch as O
    from oracle import knn_ref
    hip = deterministic_init(PMFNet(5, 3, ncls, 32, False, backbone)).cuda().eval()
    ref = deterministic_init(O.PMFNet(5, 3, ncls, 32, False, backbone)).eval()
    bs = 4
    pcd, rgb, _, mask = synthetic_batch(bs, h, w, ncls, seed=31, fill=0.3)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        rl, rc = ref(pcd, rgb)
        both = torch.cat((pcd, rgb), 1).cuda()
        lp, cp = hip(both[:, 0:5], both[:, 5:8])
    plan = next(iter(hip._plans.values()))
    assert G.rel_err(plan.read(plan.tensors["logits"]).cpu().numpy(), ref.lidar_stream.last_logits.numpy()) < 1e-3
    assert (lp.cpu() - rl).abs().max() < 1e-4 and (cp.cpu() - rc).abs().max() < 1e-4
    am_h, am_r = lp.argmax(1), rl.argmax(1)
    # argmax may legitimately differ where two classes tie to within the probability bar: the KNN vote is compared on
    # the ORACLE's argmax map (the kernel under test is the vote) and, separately, end to end with a mismatch budget
    knn = KNN({"knn": 5, "search": 5, "sigma": 1.0, "cutoff": 1.0}, ncls)
    rng = np.random.default_rng(5)
    frames, total, diff = [], 0, 0
    for b in range(bs):
        pr = np.where(mask[b].numpy() > 0, np.abs(pcd[b, 0].numpy()) + 2.0, -1.0).astype(np.float32)
        occ = np.argwhere(mask[b].numpy() > 0)
        sel = rng.integers(0, occ.shape[0], 25000)
        py, px = occ[sel, 0].astype(np.int64), occ[sel, 1].astype(np.int64)
        ur = (pr[py, px] + rng.random(sel.size).astype(np.float32) * 0.2).astype(np.float32)
        frames.append((pr, ur, px, py))
        want = knn_ref.knn_vote(pr, ur, am_r[b].numpy(), px, py)
        t = lambda a: torch.from_numpy(a).cuda()
        got = knn(t(pr), t(ur), am_r[b].cuda(), t(px), t(py)).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        e2e = knn(t(pr), t(ur), am_h[b], t(px), t(py)).cpu().numpy()
        total += want.size
        diff += int((e2e != want).sum())
    assert int((am_h.cpu() != am_r).sum()) <= 1e-4 * am_r.numel()
    assert diff <= 1e-4 * total + 2, (diff, total)
    # all frames in ONE call (batched entry point) give the same labels as frame-by-frame calls
    if hasattr(knn, "batch"):
        t = lambda a: torch.from_numpy(a).cuda()
        outs = knn.batch([(t(pr), t(ur), am_r[b].cuda(), t(px), t(py)) for b, (pr, ur, px, py) in enumerate(frames)])
        for b, (pr, ur, px, py) in enumerate(frames):
            np.testing.assert_array_equal(outs[b].cpu().numpy(), knn_ref.knn_vote(pr, ur, am_r[b].numpy(), px, py))


def test_soak_300_iterations_then_gradient_bars():
    """after hundreds of iterations on ONE batch (the bench overfits it: most gradient sums nearly cancel)
    the weight gradients of rounds 2-4 drifted past `max(3 x fp32 CPU oracle, 2e-4)` against float64 -- the fp32 accumulation
    chains of the weight-gradient kernels (4096 pixels per partial slab, fp32 fold of the slabs).  Round 5: N-split kernels
    with shorter per-wave chains and a float64 fold of the slabs.  This runs the driver's own command for 300 timed
    iterations and checks the parity block bench.py computes in the same process: logits / loss / running statistics against
    the fp32 oracle, 14 named parameter gradients (one per kernel family) against float64 with the fp32 CPU oracle's own
    distance as the yardstick (tasks/pmf/trainer.py:289-341 is the loop)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("PMF_AUTOTUNE", None)               # the bench's default: shipped table + live tuning of unknown shapes
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "300", "--warmup", "5",
                        "--no-cpu-baseline", "--no-f32-ref", "--no-roofline"], cwd=root, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    par = line["parity"]
    assert par["logits_rel"] < 1e-3 and par["loss_rel"] < 1e-4 and par["running_stat_rel"] < 1e-4, par
    bad = {k: v for k, v in par["grad_rel_vs_float64"].items() if not v["hip"] <= max(3.0 * v["cpu_fp32_oracle"], 2e-4)}
    assert not bad, bad
    assert par["ok"]


@pytest.mark.parametrize("steps,env,extra", [(3, {}, []), (1500, {}, []), (3, {"PMF_STEM_DIRECT": "1"}, []),
                                             (3, {}, ["--model", "epmf"]),
                                             (3, {}, ["--backbone", "resnet50", "--nclasses", "17", "--height", "32", "--width", "1024"])],
                         ids=["fresh", "n1500", "stem_direct", "epmf", "r50"])
def test_masked_backward_parity(steps, env, extra):
    """kinks versus defects.  The reference's backward is autograd through F.leaky_relu / F.relu /
    F.max_pool2d (salsanext.py:27-33, pmf_net.py:20-29,94; tasks/pmf/trainer.py:214-219): piecewise linear, so an activation on
    its kink moves a gradient by a whole term between two valid fp32 roundings.  Here the float64 and the
    fp32 oracle passes replay the HIP path's own decisions (Plan.act_decisions -> oracle/act_masks.py) and its upstream gradient:
    all three passes differentiate ONE piecewise-linear function, and EVERY parameter gradient of the timed plan (shipped +
    live-tuned tile table, lanes, graphs) must sit within max(3 x the fp32 oracle's distance, 2e-4) of float64 -- at the fresh
    state, at the N = 1500 bench state that failed the unmasked bars in round 5, and with the stem-class direct variant on."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("PMF_AUTOTUNE", None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", str(steps), "--warmup", "3", "--parity-masked",
                        "--no-cpu-baseline", "--no-f32-ref", "--no-roofline"] + extra, cwd=root, env=e, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    par = line["parity"]
    m = par["masked"]
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "masked_parity_%d_%s.json" % (steps, "_".join(list(env) + extra).replace("-", "") or "default")), "w") as f:
        json.dump(par, f, indent=1)
    print("[masked %d %s] params %d sites %d bad %d worst %s gmean %.2f p90 %.2f max %.2f" % (
        steps, env, m["parameters"], m["decision_sites"], m["n_bad"], m["worst"], m["ratio_gmean"], m["ratio_p90"], m["ratio_max"]))
    assert par["logits_rel"] < 1e-3
    assert m["parameters"] > 300 and m["decision_sites"] > 90
    assert m["ok"], m["bad"]
