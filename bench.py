#!/usr/bin/env python3
"""Headline benchmark: PMF-ResNet34 training iterations / second on MI355X (BASELINE.json metric, config 3).

One "step" = one full optimisation iteration of tasks/pmf/trainer.py on a synthetic, device-resident batch of
bs=2 per GPU at 64x2048 (both streams, SURVEY.md fact 4 / 8d S_A): normalise -> PMFNet forward (HIP plan) ->
focal + Lovasz (x2 heads) + perception-aware loss -> backward (HIP plan) -> AdamW(lidar) + SGD-Nesterov(camera)
-> 2 LR-scheduler steps -> confusion-matrix updates.  Dropout2d active (p=0.2), train-mode BatchNorm with local
statistics, fp32 arithmetic (fp32 MFMA).  Nothing is skipped inside the timed region.

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL); weak scaling (bs=2 per GPU); value = N*K iterations / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline     -- the conv MFMA kernel (conv_fwd_k: forward + input-gradient launches): algorithmic FLOPs of those
                  launches / their summed duration, measured with HIP events on the plan's stream in one extra
                  profiled iteration after the timed region; peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).
  cpu_baseline -- the CPU oracle (oracle/pmf_torch.py, a port pinned to reference-run fixtures) doing the same
                  iteration on the host cores (rank 0, N=1 only, one timed iteration after one warm-up).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

KITTI_MEAN = [12.12, 10.88, 0.23, -1.04, 0.21]      # tasks/pmf/config_server_kitti.yaml:80-91
KITTI_STD = [12.32, 11.47, 6.91, 0.86, 0.16]
PEAK_FP32_MFMA = 157.3                                # TFLOP/s, MI355X_MICROARCH.md


def make_batch(bs, h, w, seed, device, nclasses=20):
    from pmf_amd.utils.detinit import synthetic_batch
    pcd, rgb, label, mask = synthetic_batch(bs, h, w, nclasses, seed=seed)
    pcd = pcd * torch.tensor(KITTI_STD).view(1, 5, 1, 1) + torch.tensor(KITTI_MEAN).view(1, 5, 1, 1) * mask[:, None]
    feat = torch.cat((pcd, rgb), 1).contiguous()
    return feat.to(device), mask.to(device), label.to(device)


def infer_bench(args, model, dev):
    """BASELINE configs[1]: eval forward of bs frames + per-frame KNN vote (postproc/knn.py), everything resident."""
    from pmf_amd.postproc import KNN
    bs = 4 if args.bs == 2 else args.bs
    feat, mask, _ = make_batch(bs, args.height, args.width, 1, dev)
    model.eval()
    knn = KNN({"knn": 5, "search": 5, "sigma": 1.0, "cutoff": 1.0}, 20)
    # range image = depth channel (-1 on empty pixels), points = mask pixels x 1.3 with repeats
    frames = []
    g = torch.Generator(device="cpu").manual_seed(3)
    for b in range(bs):
        pr = torch.where(mask[b] > 0, feat[b, 0].abs() + 2.0, torch.full_like(feat[b, 0], -1.0))
        occ = torch.nonzero(mask[b] > 0)
        sel = torch.randint(0, occ.shape[0], (int(occ.shape[0] * 1.3),), generator=g).to(dev)
        py, px = occ[sel, 0].contiguous(), occ[sel, 1].contiguous()
        ur = pr[py, px] + torch.rand(sel.numel(), generator=g).to(dev) * 0.2
        frames.append((pr.contiguous(), ur.contiguous(), px, py))

    def step():
        with torch.no_grad():
            lp, _ = model(feat[:, 0:5], feat[:, 5:8])
            am = lp.argmax(1)
            return [knn(pr, ur, am[b], px, py) for b, (pr, ur, px, py) in enumerate(frames)]
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "inference frames/sec PMF-ResNet34 64x2048 bs=%d (eval forward + KNN post-processing)" % bs,
        "value": bs * args.steps / dt, "unit": "frame/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PMF-ResNet34 inference, both streams %dx%d (BASELINE configs[1]), bs=%d, KNN 5/5/1.0/1.0 "
                               "on %d points per frame" % (args.height, args.width, bs, frames[0][1].numel())},
        "roofline": None, "cpu_baseline": None}))


def salsanext_bench(args, dev, multi, rank, world):
    """the LiDAR-only task (tasks/salsanext) -- range-image loader kernels feeding one
    SalsaNextEngine iteration per step; the sweep (one per sample, 120 k points) is resident, projected inside the
    timed region.  Reported under its own metric name."""
    from oracle.cases import lidar_sweep              # synthetic input recipe only (no oracle arithmetic)
    from pmf_amd.dataset.preprocess.projection import RangeProjection
    from pmf_amd.engine import SalsaNextEngine
    from pmf_amd.models import SalsaNext
    torch.manual_seed(1)
    torch.cuda.manual_seed(1)
    model = SalsaNext(5, args.nclasses, 32).to(dev)
    eng = SalsaNextEngine(model, args.nclasses, lr=1e-3, warmup_steps=100, max_steps=15000, distributed=multi)
    rp = RangeProjection(3., -25., args.width, args.height, device=dev)
    mean, stds = torch.tensor(KITTI_MEAN, device=dev), torch.tensor(KITTI_STD, device=dev)
    sweeps = []
    for b in range(args.bs):
        pts, sem, lut = lidar_sweep(10 * rank + b, 120000)
        sweeps.append((rp.to_device(pts), torch.as_tensor(lut[sem] % args.nclasses).to(dev)))

    def step():
        items = [rp.loader_item(p, l, mean, stds) for p, l in sweeps]
        feat = torch.stack([i[0] for i in items])
        label = torch.stack([i[1] for i in items])
        mask = torch.stack([i[2] for i in items])
        return eng.train_step(feat, mask, label)
    for _ in range(args.warmup):
        step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    if not np.isfinite(float(loss)):
        raise SystemExit("bench.py: non-finite loss")
    if rank == 0:
        print(json.dumps({
            "metric": "train iters/sec SalsaNext %dx%d bs=%d (range loader + Lovasz/focal step)" % (args.height, args.width, args.bs),
            "value": world * args.steps / dt, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SalsaNext (LiDAR-only task, SURVEY 8f-2) %dx%d range image from 120k-point sweeps, "
                                   "bs=%d/GPU, AdamW" % (args.height, args.width, args.bs), "final_loss": float(loss)},
            "roofline": None, "cpu_baseline": None}))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(bs, h, w):
    """the oracle port on the host cores.  Bounded sample: the same full iteration on a 1/8-area slice of the
    workload (bs=1, H x W/4), 1 warm-up + 2 timed steps, scaled by pixel count (every term of the step is linear
    in N*H*W: convs, BN, losses; the optimiser part is size-independent and left unscaled -> slightly favours the CPU)."""
    from oracle import pmf_torch as O
    from pmf_amd.engine import TrainEngine
    torch.manual_seed(1)
    threads = min(32, os.cpu_count() or 1)      # oneDNN conv does not scale past a few dozen threads at this size
    torch.set_num_threads(threads)
    sb, sw = 1, max(w // 4, 64)
    model = O.PMFNet(5, 3, 20, 32, False, "resnet34")
    eng = TrainEngine(model, 20, feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=10, max_steps=100)
    feat, mask, label = make_batch(sb, h, sw, 1, "cpu")
    eng.train_step(feat.clone(), mask, label)
    t0 = time.time()
    n = 2
    for _ in range(n):
        eng.train_step(feat.clone(), mask, label)
    dt = (time.time() - t0) / n
    scale = (bs * h * w) / float(sb * h * sw)
    return {"value": 1.0 / (dt * scale), "unit": "iter/s", "cores": threads, "kind": "port",
            "sample": "%d timed full train iterations at bs=%d %dx%d (1/%g of the workload's pixels, %.2f s each) after "
                      "1 warm-up, scaled by pixel count; oracle/pmf_torch.py + torch %s, %d host threads"
                      % (n, sb, h, sw, scale, dt, torch.__version__, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: take the data-parallel code path (process group, range all-reduce) even with one rank")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = the headline metric; infer = BASELINE configs[1]: eval-mode forward at bs=4 + KNN "
                         "post-processing per frame, reported as frames/s under its own metric name")
    ap.add_argument("--backbone", default="resnet34", help="camera backbone (resnet50: BASELINE configs[3] family)")
    ap.add_argument("--nclasses", type=int, default=20)
    ap.add_argument("--model", default="pmf", choices=["pmf", "epmf", "salsanext"],
                    help="pmf = the headline workload (BASELINE configs[2]); epmf = configs[4] (EPMF-R34), reported "
                         "under its own metric name, no CPU baseline")
    ap.add_argument("--fresh-inputs", action="store_true",
                    help="hand the step a batch at a NEW device address every iteration (what DataLoader + .cuda() does, "
                         "tasks/pmf/trainer.py:289-303): the captured graphs must keep replaying")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-out", default=None, help="write the per-launch HIP-event profile (one line per op)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
    if args.model == "salsanext":
        return salsanext_bench(args, dev, multi, rank, world)
    from pmf_amd.engine import TrainEngine
    from pmf_amd.models import PMFNet, EPMFNet

    torch.manual_seed(1)                 # tasks/pmf/main.py:20-21: same seed on every rank
    torch.cuda.manual_seed(1)
    net = PMFNet if args.model == "pmf" else EPMFNet
    model = net(5, 3, args.nclasses, 32, imagenet_pretrained=False, image_backbone=args.backbone).to(dev)
    eng = TrainEngine(model, args.nclasses, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                      feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=10 * 100, max_steps=49 * 100,
                      distributed=multi, device_ids=[local] if multi else None)
    feat0, mask, label = make_batch(args.bs, args.height, args.width, 1 + rank, dev, args.nclasses)   # per-rank data

    ring = []

    def step():
        if args.fresh_inputs:                 # keep the last three batches alive: the allocator must rotate addresses
            batch = (feat0.clone(), mask.clone(), label.clone())
            ring.append(batch)
            if len(ring) > 3:
                ring.pop(0)
            return eng.train_step(*batch)
        return eng.train_step(feat0.clone(), mask, label)     # clone: the trainer normalises in place

    if args.mode == "infer":
        return infer_bench(args, model, dev)

    for _ in range(args.warmup):
        step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    loss_val = float(loss)
    if not np.isfinite(loss_val):
        raise SystemExit("bench.py: non-finite loss %r" % loss_val)

    roof, detail = None, None
    if rank == 0 and not args.no_roofline:
        # one extra iteration with a HIP event pair around every launch of the forward and backward plans
        plan = next(iter(model._plans.values()))
        eng.model.train()
        # rank-local measurement: no collective may be issued here (the other ranks are not taking part)
        hook = getattr(model, "_bwd_segment_hook", None)
        model._bwd_segment_hook = None
        pcd, rgb = eng.prepare(feat0.clone(), mask)
        total = eng.forward_loss(pcd, rgb, label.long())[0]
        prof_f = plan.run_profiled("forward")       # re-runs the forward plan op by op (same inputs)
        total.backward()                             # normal backward (needed to patch gradient pointers) ...
        prof_b = plan.run_profiled("backward")      # ... then the backward plan again, op by op
        model._bwd_segment_hook = hook
        if args.profile_out:
            with open(args.profile_out, "w") as f:
                for ph, prof in (("fwd", prof_f), ("bwd", prof_b)):
                    for kind, family, flops, ms, name in prof:
                        f.write("%s %-18s %-12s %-28s %9.1f us %8.2f GF %7.2f TF/s\n" % (
                            ph, kind, family or "-", name, ms * 1e3, flops / 1e9, flops / max(ms, 1e-9) / 1e9))
        fam = {}
        for kind, family, flops, ms, _ in prof_f + prof_b:
            key = family or kind
            a = fam.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += ms
        mfma_ms = fam["conv_fwd"][2] + fam["conv_dgrad"][2]
        mfma_fl = fam["conv_fwd"][1] + fam["conv_dgrad"][1]
        n_launch = fam["conv_fwd"][0] + fam["conv_dgrad"][0]
        achieved = mfma_fl / (mfma_ms * 1e-3) / 1e12
        # HBM-side bytes per launch come from a separate rocprofv3 --pmc run of this command (counters cannot be read
        # from inside the process); tools/pmc_traffic.py wrote the summary that is committed under profiles/
        traffic, tsrc = None, None
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
        headline = args.model == "pmf" and args.backbone == "resnet34" and args.nclasses == 20 and \
            (args.height, args.width, args.bs) == (64, 2048, 2)
        if headline and os.path.exists(tp):
            with open(tp) as f:
                tj = json.load(f)
            traffic, tsrc = round(tj["hbm_bytes_per_launch"]), "profiles/r01_pmc_traffic.json (" + tj["method"] + ")"
        roof = {"bound": "mfma", "kernel": "conv_fwd_k (forward + input-gradient launches, fp32 MFMA 32x32x2)",
                "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA, 4), "traffic": traffic, "traffic_source": tsrc,
                "launches_per_iter": n_launch, "avg_launch_us": round(1e3 * mfma_ms / n_launch, 2),
                "algorithmic_gflop_per_iter": round(mfma_fl / 1e9, 1)}
        detail = {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2], 3),
                      "tflops": round(v[1] / max(v[2], 1e-9) / 1e9, 2) if v[1] else None}
                  for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "pmf":
        cpu = cpu_baseline(args.bs, args.height, args.width)

    if rank == 0:
        iters = world * args.steps
        tag = ("PMF" if args.model == "pmf" else "EPMF")
        bb = {"resnet34": "ResNet34", "resnet50": "ResNet50"}.get(args.backbone, args.backbone)
        out = {
            "metric": "train iters/sec %s-%s %dx%d bs=%d/GPU (full iteration: fwd + 5-term loss + bwd + "
                      "AdamW/SGD steps)" % (tag, bb, args.height, args.width, args.bs),
            "value": iters / dt, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-%s, full train loop, both streams %dx%d (BASELINE configs[%d]%s), bs=%d/GPU, "
                                   "%d classes, dropout on, local-stat BN"
                                   % (tag, bb, args.height, args.width,
                                      4 if args.model == "epmf" else (3 if args.backbone == "resnet50" else 2),
                                      ", S_A" if (args.height, args.width) == (64, 2048) else "", args.bs,
                                      args.nclasses),
                       "global_batch": world * args.bs, "parallelism": "dp%d" % world,
                       "samples_per_s": world * args.bs * args.steps / dt, "final_loss": loss_val,
                       "fresh_input_addresses": bool(args.fresh_inputs),
                       "graphs_captured": len(next(iter(model._plans.values()))._graphs)},
            "roofline": roof, "cpu_baseline": cpu, "kernel_time_breakdown": detail,
        }
        try:       # RCCL prints its version banner through C stdio (buffered when piped): push it out first so that the
            import ctypes           # JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()             # rank 0 measured its roofline alone: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
