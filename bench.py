#!/usr/bin/env python3
"""Headline benchmark: PMF-ResNet34 training iterations / second on MI355X (BASELINE.json metric, config 3).

One "step" = one full optimisation iteration of tasks/pmf/trainer.py on a synthetic, device-resident batch of
bs=2 per GPU at 64x2048 (both streams, SURVEY.md fact 4 / 8d S_A): normalise -> PMFNet forward (HIP plan) ->
focal + Lovasz (x2 heads) + perception-aware loss -> backward (HIP plan) -> AdamW(lidar) + SGD-Nesterov(camera)
-> 2 LR-scheduler steps -> confusion-matrix updates.  Dropout2d active (p=0.2), train-mode BatchNorm with local
statistics, fp32 arithmetic (fp32 in / out / accumulate; the conv products run as six bf16 MFMA products of three-way split
fp32 operands, fp32-class error -- DESIGN.md section 4; PMF_CONV_F32=1 = v_mfma_f32_32x32x2_f32 only).  Nothing is skipped
inside the timed region.

    python bench.py --gpus N --steps K --warmup W
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL), or -- when no launcher environment (WORLD_SIZE) is present -- bench.py re-executes itself under
torch.distributed.run with N ranks on 127.0.0.1 (the reference takes its world from the launcher environment too:
pc_processor/utils/utils.py:21-44, tasks/pmf/trainer.py:33-39).  It fails loudly when fewer than N devices exist or when
the launcher's WORLD_SIZE disagrees with --gpus.  Weak scaling (bs=2 per GPU); value = N*K iterations / max-over-ranks
wall time.

Extra objects on the JSON line (every mode: --mode infer, --backbone resnet50 ..., --model epmf carry them too):
  roofline     -- the dominant kernel, the conv MFMA kernel (conv_fwd_k: forward + input-gradient launches): algorithmic
                  FLOPs of those launches / their summed duration, measured with HIP events on the plan's stream in one
                  extra op-by-op iteration after the timed region; peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).
  roofline_hbm -- the bandwidth-bound kernel families (BatchNorm backward, residual adds, pools, bilinear, softmax, fusion
                  gate, fused objective; KNN in --mode infer): SURVEY.md 8d algorithmic bytes / measured duration against
                  8 TB/s.
  cpu_baseline -- the CPU oracle (oracle/*.py, a port pinned to reference-run fixtures) doing the same step at the SAME
                  configuration on the host cores (rank 0, N=1 only: one timed iteration after one warm-up, all cores;
                  plus a single-thread number from a bounded slice).
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
ARITH_NOTE = ("fp32 in / fp32 out / fp32 accumulate; conv products of the pipelined layer class run as 6 bf16 MFMA "
              "products of 3-way split operands (error of one fp32 rounding per product; PMF_CONV_F32=1 = fp32 MFMA only)")
PEAK_BF16_MFMA = 2500.0     # TFLOP/s dense (MI355X_MICROARCH.md)
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
    knn = KNN({"knn": 5, "search": 5, "sigma": 1.0, "cutoff": 1.0}, args.nclasses)
    # range image = depth channel (-1 on empty pixels), points = mask pixels x 1.3 with repeats
    frames = []
    g = torch.Generator(device="cpu").manual_seed(3)
    for b in range(bs):
        pr = torch.where(mask[b] > 0, feat[b, 0].abs() + 2.0, torch.full_like(feat[b, 0], -1.0))
        occ = torch.nonzero(mask[b] > 0)
        sel = torch.randint(0, occ.shape[0], (int(occ.shape[0] * 1.3),), generator=g).to(dev)
        # file order of a spinning sensor: azimuth by azimuth (one firing = all lasers at one azimuth), i.e. column-major in
        # the range image -- neighbouring points of a sweep file are neighbouring pixels; --knn-random-order keeps the draw
        # order (no locality at all: the worst case for the window gather, reported beside the headline either way)
        py, px = occ[sel, 0], occ[sel, 1]
        if not getattr(args, "knn_random_order", False):
            order = torch.argsort(px * args.height + py, stable=True)
            sel, py, px = sel[order], py[order], px[order]
        py, px = py.contiguous(), px.contiguous()
        ur = pr[py, px] + torch.rand(sel.numel(), generator=g).to(dev) * 0.2
        frames.append((pr.contiguous(), ur.contiguous(), px, py))

    # all frames of the batch go through ONE KNN launch (pmf_knn_vote_batch): stacked range images, concatenated points
    pr_all = torch.stack([f[0] for f in frames])
    ur_all, px_all, py_all = (torch.cat([f[i] for f in frames]) for i in (1, 2, 3))
    counts = [f[1].numel() for f in frames]
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int64, device=dev)

    def step():
        with torch.no_grad():
            lp, _ = model(feat[:, 0:5], feat[:, 5:8])
            # class argmax (int32 map) + vote inside the library: two launches, no torch.argmax, no int64 [B, H, W] map
            return knn.forward_batch_prob(pr_all, lp, ur_all, px_all, py_all, off)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = hbm = cpu = parity = None
    if not args.no_roofline:
        plan = next(p for k, p in model._plans.items() if not k[3])
        plan.run_profiled("forward")             # (first reading discarded: see the training line's roofline block)
        roof, hbm, _ = plan_rooflines(plan, plan.run_profiled("forward"), args.model)
        am = model(feat[:, 0:5], feat[:, 5:8])[0].argmax(1)
        knn_fn, _ = knn.bind_batch(pr_all, am, ur_all, px_all, py_all, off)      # the launch alone (no wrapper overhead)
        for _ in range(200):
            knn_fn()
        ms_issue = timed_ms(knn_fn, 200)      # back-to-back launches from Python: bounded below by the ~20 us issue rate
        ms = graph_ms(knn_fn)                 # the kernel itself: 50 launches in one hipGraph
        nb = bs * 12.0 * args.height * args.width + 28.0 * ur_all.numel()    # 12 H W + 28 P bytes per frame
        pr, ur, px, py = frames[0]
        ms1 = timed_ms(lambda: knn(pr, ur, am[0], px, py), 20)
        # the same points in random order (no pixel locality between neighbouring lanes)
        perm = torch.randperm(ur_all.numel(), device=dev)
        counts_t = torch.tensor(counts, device=dev)
        fidx = torch.repeat_interleave(torch.arange(bs, device=dev), counts_t)[perm]
        perm = perm[torch.argsort(fidx, stable=True)]            # (shuffled inside every frame, frames still contiguous)
        knn_rand, _ = knn.bind_batch(pr_all, am, ur_all[perm].contiguous(), px_all[perm].contiguous(),
                                     py_all[perm].contiguous(), off)
        for _ in range(200):
            knn_rand()
        ms_rand = graph_ms(knn_rand)
        # the pair the timed step runs: in-library channel argmax (4 C H W bytes in, 4 H W out) + vote on the int32 map, against
        # torch.argmax (int64 map) + vote
        lp_now = model(feat[:, 0:5], feat[:, 5:8])[0]
        pair_fn, lab_pair = knn.forward_batch_prob(pr_all, lp_now, ur_all, px_all, py_all, off, bind=True)
        ms_pair = graph_ms(pair_fn)
        try:
            ms_torch_pair = graph_ms(lambda: knn.forward_batch(pr_all, lp_now.argmax(1), ur_all, px_all, py_all, off), reps=10)
        except Exception:       # (an allocation inside the capture the caching allocator refuses)
            ms_torch_pair = timed_ms(lambda: knn.forward_batch(pr_all, lp_now.argmax(1), ur_all, px_all, py_all, off), 50)
        assert torch.equal(lab_pair, knn.forward_batch(pr_all, lp_now.argmax(1), ur_all, px_all, py_all, off))
        nb_am = 4.0 * lp_now.numel() + 4.0 * pr_all.numel()
        hbm.insert(0, {"kernel": "knn_batch_lds_k (5x5 window staged through LDS per 256-point workgroup, k=5 vote per point; "
                                 "all %d frames in one launch; launch time = HIP events around a hipGraph of 50 launches)" % bs,
                       "bound": "hbm", "launches": 1, "achieved": round(nb / ms / 1e6, 1), "peak": PEAK_HBM, "unit": "GB/s",
                       "frac": round(nb / ms / 1e6 / PEAK_HBM, 5), "algorithmic_mb_per_iter": round(nb / 1e6, 2),
                       "ms_per_iter": round(ms, 4), "points": int(ur_all.numel()),
                       "frac_of_launch_floor": round(max(nb / 6.3e12 * 1e3, 4e-3) / ms, 4),
                       "per_frame_launch_us": round(1e3 * ms1, 2), "python_issue_loop_us": round(1e3 * ms_issue, 2),
                       "point_order": "random draw order" if args.knn_random_order else "sweep-file order (azimuth-major)",
                       "random_point_order_us": round(1e3 * ms_rand, 2),
                       "argmax_plus_vote": {"what": "pmf_knn_vote_batch_prob: argmax_nchw_k (int32 label map) + the vote, what the "
                                                    "timed step runs", "us": round(1e3 * ms_pair, 2),
                                            "torch_argmax_plus_vote_us": round(1e3 * ms_torch_pair, 2) if ms_torch_pair else None,
                                            "argmax_algorithmic_mb": round(nb_am / 1e6, 2),
                                            "argmax_gbps_upper_bound_of_pair": round(nb_am / max(ms_pair - ms, 1e-6) / 1e6, 1)}})
    if not args.no_parity:
        parity = infer_parity(args, model, feat, mask, frames, knn, out.split(counts))
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(bs, args.height, args.width, args.model, args.backbone, args.nclasses, mode="infer")
    print(json.dumps({
        "metric": "inference frames/sec PMF-%s %dx%d bs=%d (eval forward + KNN post-processing)" % (
            {"resnet34": "ResNet34", "resnet50": "ResNet50"}.get(args.backbone, args.backbone), args.height, args.width, bs),
        "value": bs * args.steps / dt, "unit": "frame/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "arithmetic": ARITH_NOTE,
        "config": {"workload": "PMF-%s inference, both streams %dx%d (BASELINE configs[1]%s), bs=%d, %d classes, KNN 5/5/1.0/1.0 "
                               "on %d points per frame" % ({"resnet34": "ResNet34", "resnet50": "ResNet50"}.get(args.backbone, args.backbone),
                                                           args.height, args.width,
                                                           "" if args.backbone == "resnet34" else "'s loop on configs[3]'s network",
                                                           bs, args.nclasses, frames[0][1].numel())},
        "parity": parity, "roofline": roof, "roofline_hbm": hbm, "cpu_baseline": cpu}))


def infer_parity(args, model, feat, mask, frames, knn, labels_timed):
    """configs[1] parity, in the process that was timed: the eval forward of all bs frames against the CPU oracle
    (pre-softmax logits rel, bar 1e-3; probabilities abs, bar 1e-4), the KNN labels of EVERY frame against
    oracle/knn_ref.py on the oracle's own argmax map (bit-exact), and end to end (labels of the timed step vs the oracle
    chain; a pixel whose two best classes tie to within the probability bar may flip its argmax: mismatch budget 1e-4)."""
    from oracle import knn_ref
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    net = _oracle_model(args.model, args.backbone, args.nclasses)
    net.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    net.eval()
    f = feat.detach().cpu()
    with torch.no_grad():
        rl, rc = net(f[:, 0:5], f[:, 5:8])
        lp, cp = model(feat[:, 0:5], feat[:, 5:8])
    plan = next(p for k, p in model._plans.items() if not k[3])
    logits = plan.read(plan.tensors["logits"]).float().cpu()
    ref_logits = net.lidar_stream.last_logits.detach()
    lrel = float(((logits - ref_logits).abs() / ref_logits.abs().clamp_min(1.0)).max())
    pabs = max(float((lp.cpu() - rl).abs().max()), float((cp.cpu() - rc).abs().max()))
    am_r = rl.argmax(1)
    exact, e2e_diff, total = True, 0, 0
    for b, (pr, ur, px, py) in enumerate(frames):
        want = knn_ref.knn_vote(pr.cpu().numpy(), ur.cpu().numpy(), am_r[b].numpy(), px.cpu().numpy(), py.cpu().numpy())
        got = knn(pr, ur, am_r[b].to(pr.device), px, py).cpu().numpy()
        exact = exact and bool(np.array_equal(got, want))
        e2e_diff += int((labels_timed[b].cpu().numpy() != want).sum())
        total += want.size
    return {"logits_rel": lrel, "prob_abs": pabs, "knn_labels_exact_on_oracle_argmax": exact,
            "end_to_end_label_mismatch": e2e_diff, "points": total,
            "bars": {"logits_rel": 1e-3, "prob_abs": 1e-4, "end_to_end_label_mismatch_frac": 1e-4},
            "ok": bool(lrel < 1e-3 and pabs < 1e-4 and exact and e2e_diff <= 1e-4 * total + 2),
            "what": "eval forward of the %d timed frames + KNN of every frame against oracle/pmf_torch.py + oracle/knn_ref.py "
                    "(%d host threads), same weights, same inputs" % (len(frames), torch.get_num_threads())}


def loader_bench(args, dev):
    """the perspective-projection loader, reported separately from the training step.
    Part A -- one KITTI-sized frame (120 k raw LiDAR points, a 376 x 1241 camera image): microseconds per frame of every
    loader kernel (HIP events, launches back to back), algorithmic bytes of SURVEY 8d (projection 16 P + 43 h w) against
    8 TB/s, the numpy oracle (oracle/loader_ref.py, one core) beside it.
    Part B -- loader in the loop: a synthetic on-disk SemanticKITTI tree (oracle/cases.kitti_tree: .bin / .label / .png /
    calib.txt) read by the SemanticKitti parser, projected / augmented by PerspectiveViewLoader exactly as
    tasks/pmf/trainer.py builds it (is_train, img_aug, use_padding; batches of 2 through DataLoader + the prefetch
    thread) feeding TrainEngine.train_step at the KITTI training crop 256 x 1024 (the largest crop a 376 x 1241 frame
    yields; 64 x 2048 cannot be cut from it), next to the same step on a resident batch of the same shape."""
    import shutil
    import tempfile
    import yaml
    from oracle import loader_ref
    from oracle.cases import lidar_sweep, kitti_tree
    from pmf_amd.dataset import perspective_view_loader as PV
    from pmf_amd.dataset.semantic_kitti import SemanticKitti
    from pmf_amd.engine import TrainEngine
    from pmf_amd.models import PMFNet
    from pmf_amd.utils.detinit import deterministic_init
    from tasks.pmf.trainer import Prefetcher
    from torch.utils.data import DataLoader
    H, W, P = 376, 1241, 120000
    pts, sem, lut = lidar_sweep(5, P)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    P2 = np.array([[0.58 * W, 0, W / 2.0, 4.5e1], [0, 0.58 * W, H / 2.0, -0.3], [0, 0, 1.0, 2.7e-3]])
    Tr = np.array([[4.2e-4, -9.9996e-1, -8.4e-3, -1.2e-2], [-7.2e-3, 8.4e-3, -9.9993e-1, -5.4e-2],
                   [9.9997e-1, 4.8e-4, -7.2e-3, -2.9e-1], [0, 0, 0, 1.0]])
    mat = P2 @ Tr
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tasks", "pmf", "config_server_kitti.yaml")))
    sensor = cfg["sensor"]
    d_pts = torch.from_numpy(pts).to(dev)
    d_img = torch.from_numpy(img).to(dev)

    def ev_ms(fn, reps=50):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    proj, xd, yd, depth, keep = PV.project_frame_gpu(d_pts, sem, d_img, mat, lut, dev)
    kept = int(xd.numel())
    rows = []

    def add(name, ms, nbytes, note="", launches=None):
        row = {"kernel": name, "us_per_frame": round(1e3 * ms, 2), "algorithmic_mb": round(nbytes / 1e6, 3),
               "achieved": round(nbytes / ms / 1e6, 1), "peak": PEAK_HBM, "unit": "GB/s",
               "frac": round(nbytes / ms / 1e6 / PEAK_HBM, 5), "note": note}
        if launches:    # the same yardstick as plan_rooflines: bytes at the achievable copy rate, 4 us per dependent launch
            row["launches"] = launches
            row["frac_of_launch_floor"] = round(max(nbytes / 6.3e12 * 1e3, 4e-3 * launches) / ms, 4)
        rows.append(row)
    # as PerspectiveViewLoader runs it: points / labels / image arrive in ONE packed upload (upload_packed) BEFORE the call,
    # the calibration matrix and the label LUT are per-sequence device constants -- so all of them are resident here; the
    # wrapper's output allocations are inside the timed call
    d_sem = torch.from_numpy(np.ascontiguousarray(sem, np.int32)).to(dev)
    d_mat = torch.from_numpy(np.ascontiguousarray(mat, np.float64).reshape(12)).to(dev)
    d_lut = torch.from_numpy(np.ascontiguousarray(lut, np.int32)).to(dev)
    add("pmf_project_scatter2 (project + ordered compaction + scatter in one launch, gather in a second; via project_frame_gpu)",
        ev_ms(lambda: PV.project_frame_gpu(d_pts, d_sem, d_img, d_mat, d_lut, dev, need_uproj=False)),
        16.0 * P + 43.0 * H * W, launches=2, note="P = %d raw points, %d inside the image; inputs resident (the loader's one packed upload "
        "precedes the call); includes the wrapper's output allocations; the five-launch form with per-call uploads of "
        "labels / matrix / LUT measured 110-126 us" % (P, kept))
    add("pmf_crop_pad (validation: CenterCrop + Pad, 10 channels)",
        ev_ms(lambda: PV.center_crop_pad_gpu(proj, sensor["proj_h"], sensor["proj_w"], sensor["h_pad"], sensor["w_pad"])),
        4.0 * 10 * (min(H, sensor["proj_h"]) * min(W, sensor["proj_w"]) + sensor["proj_h"] * sensor["proj_w"]))
    frc = PV.FlipRotateCrop(sensor["proj_ht"] - 2 * sensor["h_pad"], sensor["proj_wt"] - 2 * sensor["w_pad"],
                            sensor["h_pad"], sensor["w_pad"])
    add("pmf_flip_rotate_crop (training: flip + rotation 15 deg + RandomCrop + Pad as one gather, 10 channels)",
        ev_ms(lambda: frc.apply(proj, True, 9.0, 40, 100)), 4.0 * 10 * 2 * sensor["proj_ht"] * sensor["proj_wt"])
    cj = PV.ColorJitter(*cfg["augmentation"]["img_jitter"])
    scratch = d_img.clone()
    add("pmf_color_jitter (training: brightness, contrast, saturation on the uint8 frame)",
        ev_ms(lambda: cj.apply(scratch, [0, 1, 2, 3], [1.2, 0.8, 1.1, None])), 3.0 * H * W * 2 * 3,
        "three active operations (hue range is zero in config_server_kitti.yaml), each one read + one write of the frame")
    # numpy oracle, one core, same frame
    t0 = time.time()
    rp, rx, ry, rd = loader_ref.project_frame(pts, sem, img, mat, lut)
    loader_ref.center_crop_pad(rp, sensor["proj_h"], sensor["proj_w"], sensor["h_pad"], sensor["w_pad"])
    t_np = time.time() - t0
    same = bool(np.array_equal(proj.cpu().numpy(), rp))
    gpu_eval_us = rows[0]["us_per_frame"] + rows[1]["us_per_frame"]

    # ---- part B: loader in the loop
    root = tempfile.mkdtemp(prefix="pmf_kitti_")
    loop = None
    try:
        nfr = 12
        cfg_path, _ = kitti_tree(root, seed=1, seqs=(0, 8), frames=nfr, npts=P, h=H, w=W)
        ds = SemanticKitti(root=root, sequences=[0], config_path=cfg_path)
        ncls = len(ds.mapped_cls_name) if hasattr(ds, "mapped_cls_name") else 20
        cfg["augmentation"]["img_jitter"] = cfg["augmentation"]["img_jitter"]
        pv = PV.PerspectiveViewLoader(dataset=ds, config=cfg, is_train=True, pcd_aug=False, img_aug=True, use_padding=True,
                                      device=dev)
        # host part alone (file read + PNG decode), per frame
        t0 = time.time()
        for i in range(len(ds)):
            ds.loadDataByIndex(i)
            np.array(ds.loadImage(i))
        host_ms = 1e3 * (time.time() - t0) / len(ds)
        t0 = time.time()
        for i in range(len(ds)):
            pv[i]
        torch.cuda.synchronize()
        item_ms = 1e3 * (time.time() - t0) / len(ds)
        torch.manual_seed(1)
        model = deterministic_init(PMFNet(5, 3, 20, 32, imagenet_pretrained=False, image_backbone="resnet34")).to(dev)
        eng = TrainEngine(model, 20, lr=1e-3, feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=1000, max_steps=4900)
        hh, ww = sensor["proj_ht"], sensor["proj_wt"]
        feat0, mask0, label0 = make_batch(2, hh, ww, 1, dev, 20)

        def run_resident(k):
            for _ in range(k):
                eng.train_step(feat0.clone(), mask0, label0)
        run_resident(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_resident(args.steps)
        torch.cuda.synchronize()
        res_ips = args.steps / (time.perf_counter() - t0)

        def batches(workers):
            while True:
                dl = DataLoader(pv, batch_size=2, num_workers=0, shuffle=True, drop_last=True)
                for b in (Prefetcher(dl, workers=workers) if workers else dl):
                    yield b

        def timed_loop(workers, train):
            it = batches(workers)
            for _ in range(4):
                f, m, l = next(it)
                if train:
                    eng.train_step(f, m, (l % 20))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                f, m, l = next(it)
                if train:
                    eng.train_step(f, m, (l % 20))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            it.close()
            return args.steps / dt
        out = {"in_line": timed_loop(0, True), "prefetch_1": timed_loop(1, True), "prefetch_2": timed_loop(2, True),
               "pipeline_alone_1": 2 * timed_loop(1, False), "pipeline_alone_2": 2 * timed_loop(2, False),
               "pipeline_alone_4": 2 * timed_loop(4, False)}
        loop = {"workload": "PMF-ResNet34 train step at the KITTI training crop %dx%d, bs=2, batches from a synthetic "
                            "on-disk SemanticKITTI tree (%d frames of %d points + %dx%d PNG)" % (hh, ww, len(ds), P, H, W),
                "resident_batch_iter_per_s": res_ips,
                "loader_in_loop_iter_per_s": {"no prefetch (loader calls on the training stream)": out["in_line"],
                                              "1 prefetch thread": out["prefetch_1"], "2 prefetch threads": out["prefetch_2"]},
                "loader_pipeline_alone_frames_per_s": {"1 thread": out["pipeline_alone_1"], "2 threads": out["pipeline_alone_2"],
                                                       "4 threads": out["pipeline_alone_4"]},
                "host_read_decode_ms_per_frame": host_ms, "loader_item_ms_per_frame": item_ms,
                "frames_per_s_consumed": {"this crop, resident batch": 2 * res_ips,
                                          "headline step (64x2048, ~55 it/s)": 110.0},
                "steps": args.steps}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print(json.dumps({
        "metric": "perspective loader: frames/sec of the eval path (project + crop/pad) on one KITTI-sized frame",
        "value": 1e6 / gpu_eval_us, "unit": "frame/s", "n_gpus": 1, "steps": 50, "warmup": 10,
        "ms_per_step": gpu_eval_us / 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 projection / int32 indices / f32 features", "data": "synthetic",
        "config": {"workload": "PerspectiveViewLoader kernels on a %dx%d frame with %d raw points (%d in the image); "
                               "sensor block of tasks/pmf/config_server_kitti.yaml" % (H, W, P, kept),
                   "bit_exact_vs_numpy_oracle": same},
        "roofline": rows[0] | {"bound": "hbm"}, "roofline_hbm": rows,
        "cpu_baseline": {"value": 1.0 / t_np, "unit": "frame/s", "cores": 1, "kind": "port",
                         "sample": "one frame, project_frame + center_crop_pad of oracle/loader_ref.py (numpy): %.1f ms" % (1e3 * t_np)},
        "loader_in_loop": loop}))


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
        return eng.train_step(feat, label, mask)
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
            "arithmetic": ARITH_NOTE,
            "config": {"workload": "SalsaNext (LiDAR-only task, SURVEY 8f-2) %dx%d range image from 120k-point sweeps, "
                                   "bs=%d/GPU, AdamW" % (args.height, args.width, args.bs), "final_loss": float(loss)},
            "roofline": None, "cpu_baseline": None}))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def _oracle_model(model, backbone, nclasses):
    from oracle import pmf_torch as O
    if model == "epmf":
        from oracle import epmf_torch as E
        return E.EPMFNet(5, 3, nclasses, 32, False, backbone)
    return O.PMFNet(5, 3, nclasses, 32, False, backbone)


def cpu_baseline(bs, h, w, model="pmf", backbone="resnet34", nclasses=20, mode="train"):
    """the CPU oracle (oracle/*.py: the port pinned to reference-run fixtures) doing the SAME step at the SAME
    configuration on the host cores: 1 warm-up + 1 timed iteration with every core torch will use, plus -- for scaling
    context -- one single-thread iteration of a 1/16-area slice (bs=1, W/8) scaled by pixel count."""
    from pmf_amd.engine import TrainEngine, EPMFEngine
    Engine = EPMFEngine if model == "epmf" else TrainEngine
    torch.manual_seed(1)
    # oneDNN's fp32 convolutions stop scaling at a few dozen threads at these sizes (256 threads on the GPU box's host ran
    # the same pass 20x SLOWER than 32): the baseline uses the count that is fastest, and says so
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    net = _oracle_model(model, backbone, nclasses)
    feat, mask, label = make_batch(bs, h, w, 1, "cpu", nclasses)
    if mode == "infer":
        from oracle import knn_ref
        net.eval()
        pr = torch.where(mask[0] > 0, feat[0, 0].abs() + 2.0, torch.full_like(feat[0, 0], -1.0)).numpy()
        occ = np.argwhere(mask[0].numpy() > 0)
        sel = np.random.default_rng(3).integers(0, occ.shape[0], int(occ.shape[0] * 1.3))
        py, px = occ[sel, 0].astype(np.int64), occ[sel, 1].astype(np.int64)
        ur = (pr[py, px] + 0.1).astype(np.float32)

        def fwd():
            with torch.no_grad():
                lp, _ = net(feat[:, 0:5], feat[:, 5:8])
                return lp.argmax(1).numpy()
        fwd()
        t0 = time.time()
        am = fwd()
        dt_f = time.time() - t0
        t0 = time.time()
        knn_ref.knn_vote(pr, ur, am[0], px, py)          # one frame of the bs (numpy, one core), scaled by bs
        dt_k = (time.time() - t0) * bs
        dt = dt_f + dt_k
        return {"value": bs / dt, "unit": "frame/s", "cores": threads, "kind": "port",
                "sample": "full configuration: 1 timed eval forward of bs=%d at %dx%d after 1 warm-up (%.2f s, %d of %d host "
                          "threads) + the KNN vote of ONE frame (%d points) scaled by bs (%.2f s, numpy, 1 core); "
                          "oracle/pmf_torch.py + oracle/knn_ref.py, torch %s"
                          % (bs, h, w, dt_f, threads, os.cpu_count() or 1, ur.shape[0], dt_k, torch.__version__)}
    eng = Engine(net, nclasses, feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=10, max_steps=100)
    # warm-up: one iteration on a 1/8-area slice (thread pools) and one at the full size (oneDNN primitive caches for these
    # shapes), then THREE timed iterations at the full size (warm-up + several); the median is reported
    fw, mw, lw = make_batch(1, h, max(w // 4, 64), 2, "cpu", nclasses)
    eng.train_step(fw.clone(), mw, lw)
    eng.train_step(feat.clone(), mask, label)
    times = []
    for _ in range(3):
        t0 = time.time()
        eng.train_step(feat.clone(), mask, label)
        times.append(time.time() - t0)
    dt = sorted(times)[1]
    # single thread, bounded: 1/16 of the pixels
    torch.set_num_threads(1)
    sw = max(w // 8, 64)
    f1, m1, l1 = make_batch(1, h, sw, 1, "cpu", nclasses)
    net1 = _oracle_model(model, backbone, nclasses)
    eng1 = Engine(net1, nclasses, feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=10, max_steps=100)
    t1 = time.time()
    eng1.train_step(f1.clone(), m1, l1)
    d1 = (time.time() - t1) * (bs * h * w) / float(h * sw)
    torch.set_num_threads(threads)
    return {"value": 1.0 / dt, "unit": "iter/s", "cores": threads, "kind": "port",
            "sample": "median of 3 timed full train iterations at the full configuration bs=%d %dx%d (%s s; after one warm-up "
                      "iteration on a 1/8-area slice and one at the full size); oracle + torch %s on %d of %d host threads "
                      "(more threads are slower)"
                      % (bs, h, w, " / ".join("%.2f" % t for t in times), torch.__version__, threads, os.cpu_count() or 1),
            "single_thread": {"value": 1.0 / d1, "unit": "iter/s", "cores": 1,
                              "sample": "1 iteration at bs=1 %dx%d on one thread, scaled by pixel count to the full "
                                        "configuration (%.1f s)" % (h, sw, d1)}}


def _detach_dp_hooks(model):
    """rank-local measurements must not issue collectives: take the data-parallel hooks off the model, return them"""
    saved = {k: getattr(model, k, None) for k in ("_bwd_segment_hook", "_bwd_gated_hook")}
    for k in saved:
        setattr(model, k, None)
    return saved


def _attach_dp_hooks(model, saved):
    for k, v in saved.items():
        setattr(model, k, v)


def parity_block(args, eng, model, feat0, mask, label):
    """The plan that was just timed (autotuned tile configurations, lanes, captured hipGraphs) against the CPU oracle, in
    this process: the model's CURRENT state (after the warm-up and timed iterations) is copied into the oracle, both get
    the same Dropout2d multipliers, and one train-mode forward + objective runs on each side from the same batch.
    Reported: pre-softmax LiDAR logits (max |d| / max(|ref|, 1), bar 1e-3 = BASELINE.json), total loss (relative, bar
    1e-4), BatchNorm running statistics after the pass (max |d| relative to the tensor's scale, bar 1e-4)."""
    from oracle import pmf_torch as O
    from pmf_amd.engine import TrainEngine, EPMFEngine
    dev = feat0.device
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    net = _oracle_model(args.model, args.backbone, args.nclasses)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    net.load_state_dict(sd)
    net.train()
    Engine = EPMFEngine if args.model == "epmf" else TrainEngine
    ref_eng = Engine(net, args.nclasses, lambda_=1.0, gamma=0.5, tau=0.7, feature_mean=KITTI_MEAN, feature_std=KITTI_STD,
                     warmup_steps=10, max_steps=100)
    if args.model == "epmf":
        with torch.no_grad():
            ref_eng.mt_loss.sigma.copy_(eng.mt_loss.sigma.detach().cpu())
    g = torch.Generator().manual_seed(3)
    masks = {n: (torch.rand(args.bs, c, generator=g) > 0.2).float() / 0.8 for n, c in model._mask_sites()}
    O.set_dropout_masks(net, masks)
    model.set_dropout_masks({k: v.to(dev) for k, v in masks.items()})
    try:
        eng.model.train()
        hooks = _detach_dp_hooks(model)     # (world 1 only: no collective inside this backward)
        pcd, rgb = eng.prepare(feat0.clone(), mask)
        total, _, lp_h, cp_h, _ = eng.forward_loss(pcd, rgb, label.long())
        lp_h.retain_grad()
        cp_h.retain_grad()
        total.backward()                    # the timed backward plan (flat state: writes every p.grad in place)
        torch.cuda.synchronize()
        gobj_h = (lp_h.grad.detach().cpu(), cp_h.grad.detach().cpu())     # d objective / d probability maps, as the HIP path saw it
        plan = next(p for k, p in model._plans.items() if k[3])
        logits = plan.read(plan.tensors["logits"]).float().cpu()
        loss_h = float(total.detach())
        rs_h = {k: v.detach().cpu() for k, v in model.state_dict().items() if "running_" in k}
        named = dict(model.named_parameters())
        picks = [k for k in PARITY_GRADS if k in named and named[k].grad is not None]
        grads_h = {k: named[k].grad.detach().float().cpu().clone() for k in picks}
        graphs = len(plan._graphs)
        if getattr(args, "parity_masked", False):
            decisions = plan.act_decisions(model)       # sign masks / argmax positions of THIS forward pass
            grads_all = {k: p.grad.detach().float().cpu().clone() for k, p in named.items() if p.grad is not None}
    finally:
        model.set_dropout_masks(None)
        _attach_dp_hooks(model, hooks)
    # The objective is DISCONTINUOUS (confidence thresholds of the perception-aware terms, the Lovasz ranking): in a long-trained
    # state a rounding-level difference in the probabilities moves its gradient by whole terms -- measured (round 5,
    # tools/soak_tensors.py): BOTH fp32 paths sit 1.1-1.3e-3 from float64 there, in 17 elements beyond 1e-3 of the maximum, and
    # which elements flip differs from path to path; one such flip shifts every LiDAR-stream gradient by the same factor while
    # all forward tensors agree to 1.2x.  So the two things are reported separately: (1) the objective's gradient w.r.t. the two
    # probability maps, each path against float64; (2) the NETWORK's backward pass given ONE upstream gradient -- the oracle
    # passes below are driven by the gradient the HIP path computed, so that (2) measures the kernels, not the objective's kinks.
    def oracle_backward(e_, net_, dt, inject=None):
        p_, r_ = e_.prepare(feat0.detach().cpu().to(dt), mask.cpu().to(dt))
        if inject is not None:
            from oracle.act_masks import ActSites
            with ActSites(net_, inject=inject):
                tot_, _, lp_o, cp_o, _ = e_.forward_loss(p_, r_, label.cpu().long())
        else:
            tot_, _, lp_o, cp_o, _ = e_.forward_loss(p_, r_, label.cpu().long())
        own = torch.autograd.grad(tot_, [lp_o, cp_o], retain_graph=True)        # the oracle's own objective gradient
        torch.autograd.backward([lp_o, cp_o], [gobj_h[0].to(dt), gobj_h[1].to(dt)])
        return float(tot_.detach()), [g.detach().double() for g in own]
    loss_r, gobj_32 = oracle_backward(ref_eng, net, torch.float32)             # tasks/pmf/trainer.py:214-219 on the CPU oracle, fp32
    ref_named = dict(net.named_parameters())
    # ... and the same pass in FLOAT64: two fp32 paths with different rounding points are each ~1e-2 away from the exact
    # gradient in this network (the BatchNorm backward subtracts two per-channel means from gy in ~90 layers; measured in
    # tests/test_gpu_fullsize.py), so "HIP vs fp32 oracle" alone cannot tell a rounding difference from a defect -- the
    # yardstick of a parameter is the fp32 CPU oracle's own distance from float64
    net64 = _oracle_model(args.model, args.backbone, args.nclasses)
    net64.load_state_dict(sd)                  # (the state before the fp32 pass above touched the running statistics)
    net64 = net64.double().train()
    O.set_dropout_masks(net64, {k: v.double() for k, v in masks.items()})
    eng64 = Engine(net64, args.nclasses, lambda_=1.0, gamma=0.5, tau=0.7, feature_mean=KITTI_MEAN, feature_std=KITTI_STD,
                   warmup_steps=10, max_steps=100)
    eng64.focal.double()
    if args.model == "epmf":
        eng64.mt_loss.double()
        with torch.no_grad():
            eng64.mt_loss.sigma.copy_(eng.mt_loss.sigma.detach().cpu().double())
    _, gobj_64 = oracle_backward(eng64, net64, torch.float64)
    named64 = dict(net64.named_parameters())
    obj = {}
    for nm, i in (("lidar", 0), ("camera", 1)):
        den = gobj_64[i].norm().clamp_min(1e-30)
        thr = 1e-3 * float(gobj_64[i].abs().max())
        obj[nm] = {"hip": float((gobj_h[i].double() - gobj_64[i]).norm() / den),
                   "cpu_fp32_oracle": float((gobj_32[i] - gobj_64[i]).norm() / den),
                   "elements_beyond_1e-3_of_max": {"hip": int(((gobj_h[i].double() - gobj_64[i]).abs() > thr).sum()),
                                                   "cpu_fp32_oracle": int(((gobj_32[i] - gobj_64[i]).abs() > thr).sum())}}
    grad_rel, grad_rel_cpu = {}, {}
    for k in picks:
        g64 = named64[k].grad.detach()
        den = g64.norm().clamp_min(1e-30)
        grad_rel[k] = float((grads_h[k].double() - g64).norm() / den)
        grad_rel_cpu[k] = float((ref_named[k].grad.detach().double() - g64).norm() / den)
    masked = None
    if getattr(args, "parity_masked", False):
        masked = _masked_backward(args, Engine, eng, sd, masks, decisions, grads_all, oracle_backward)
    ref_logits = net.lidar_stream.last_logits.detach()
    if logits.shape != ref_logits.shape:                       # the plan stores NHWC
        logits = logits.permute(0, 3, 1, 2)[:, :ref_logits.shape[1]]
    lrel = float(((logits - ref_logits).abs() / ref_logits.abs().clamp_min(1.0)).max())
    # the exact answer is the float64 oracle: where the reference's own fp32 CPU path is the outlier (PMF-ResNet50 at 2 x 480 x 640:
    # 0.9e-3 from float64), the HIP path is held to max(1e-3, 2 x the fp32 oracle's own distance) from float64 instead
    l64 = net64.lidar_stream.last_logits.detach()
    lrel64 = float(((logits.double() - l64).abs() / l64.abs().clamp_min(1.0)).max())
    lrel64_cpu = float(((ref_logits.double() - l64).abs() / l64.abs().clamp_min(1.0)).max())
    logits_ok = lrel < 1e-3 or lrel64 <= max(1e-3, 2.0 * lrel64_cpu)
    rrel = 0.0
    for k, v in net.state_dict().items():
        if "running_" in k:
            rrel = max(rrel, float((rs_h[k] - v).abs().max() / max(float(v.abs().max()), 1.0)))
    lossrel = abs(loss_h - loss_r) / max(abs(loss_r), 1.0)
    gratio = max(grad_rel[k] / max(grad_rel_cpu[k], 1e-12) for k in picks if max(grad_rel[k], grad_rel_cpu[k]) > 2e-4) \
        if any(max(grad_rel[k], grad_rel_cpu[k]) > 2e-4 for k in picks) else 0.0
    gok = all(grad_rel[k] <= max(3.0 * grad_rel_cpu[k], 2e-4) for k in picks)
    # (a NaN in the fp32 oracle's own objective gradient -- torch's xlogy / KLDiv backward at a probability that is exactly 0 in
    # fp32, seen with EPMF at 320 x 1280 -- is no yardstick: the HIP path is then held to the floor alone)
    ook = all(v["hip"] <= max(3.0 * (0.0 if v["cpu_fp32_oracle"] != v["cpu_fp32_oracle"] else v["cpu_fp32_oracle"]), 2e-4)
              for v in obj.values())
    return {"logits_rel": lrel, "logits_rel_vs_float64": {"hip": lrel64, "cpu_fp32_oracle": lrel64_cpu},
            "loss_rel": lossrel, "running_stat_rel": rrel,
            "objective_grad_rel_vs_float64": obj,
            "grad_rel_vs_float64": {k: {"hip": grad_rel[k], "cpu_fp32_oracle": grad_rel_cpu[k]} for k in picks},
            "grad_rel_worst": max(grad_rel.values()) if picks else None,
            "grad_rel_worst_ratio_to_cpu_fp32": gratio,
            "loss_hip": loss_h, "loss_oracle": loss_r, "masked": masked,
            "bars": {"logits_rel": "hip vs fp32 oracle < 1e-3, or hip vs float64 <= max(1e-3, 2 x fp32 oracle vs float64)",
                     "loss_rel": 1e-4, "running_stat_rel": 1e-4,
                     "grad_rel": "hip <= max(3 x cpu_fp32_oracle, 2e-4) for every listed parameter, all three backward passes "
                                 "driven by the SAME upstream gradient (the HIP path's d objective / d probabilities)",
                     "objective_grad_rel": "hip <= max(3 x cpu_fp32_oracle, 2e-4) per probability map, each path's own objective"},
            "ok": bool(logits_ok and lossrel < 1e-4 and rrel < 1e-4 and gok and ook and (masked is None or masked["ok"])),
            "what": "train-mode forward + objective + BACKWARD of the plan that was timed (PMF_AUTOTUNE %s, lanes, %d captured "
                    "graphs) against the CPU oracle: model state after the timed iterations copied into oracle/ (%d host "
                    "threads), same Dropout2d multipliers, same batch; logits / loss / running statistics against the fp32 "
                    "oracle; grad_rel_vs_float64 = |g - g64|_2 / |g64|_2 of the named parameter gradients (first / last "
                    "layer of each stream, one layer per kernel family) for the HIP path and for the fp32 CPU oracle, both "
                    "against the oracle in float64" % (os.environ.get("PMF_AUTOTUNE", "on"), graphs, torch.get_num_threads())}


def _masked_backward(args, Engine, eng, sd, masks, decisions, grads_h, oracle_backward):
    """--parity-masked: the float64 and the fp32 oracle passes once more, now with the HIP path's
    piecewise-linear DECISIONS injected (sign of every LeakyReLU / ReLU pre-activation, the stem max-pool's argmax;
    oracle/act_masks.py, Plan.act_decisions) and, as before, driven by the HIP path's d objective / d probabilities.  All three
    backward passes then differentiate the SAME piecewise-linear function (salsanext.py:27-33, pmf_net.py:20-29,94 are the
    reference's activations): an activation sitting on its kink can no longer move a gradient by a whole term, so what is left
    between the passes is rounding -- or a kernel defect.  EVERY parameter of the flat state is compared."""
    from oracle import pmf_torch as O
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        net_ = _oracle_model(args.model, args.backbone, args.nclasses)
        net_.load_state_dict(sd)
        net_ = net_.to(dt).train()
        O.set_dropout_masks(net_, {k: v.to(dt) for k, v in masks.items()})
        e_ = Engine(net_, args.nclasses, lambda_=1.0, gamma=0.5, tau=0.7, feature_mean=KITTI_MEAN, feature_std=KITTI_STD,
                    warmup_steps=10, max_steps=100)
        if dt == torch.float64:
            e_.focal.double()
        if args.model == "epmf":
            e_.mt_loss.to(dt)
            with torch.no_grad():
                e_.mt_loss.sigma.copy_(eng.mt_loss.sigma.detach().cpu().to(dt))
        oracle_backward(e_, net_, dt, inject=decisions)
        out[tag] = {k: p.grad.detach().double() for k, p in net_.named_parameters() if p.grad is not None}
        del net_, e_
    rows = []
    for k, gh in grads_h.items():
        if k not in out["f64"]:
            continue
        g64 = out["f64"][k]
        wk = k.rsplit(".", 1)[0] + ".weight"
        # (a conv bias in front of a train-mode BatchNorm has a true gradient of exactly 0: its layer's weight gradient sets
        # the scale, as in tests/test_gpu_fullsize.py)
        floor = 1e-6 * float(out["f64"][wk].norm()) if wk in out["f64"] else 0.0
        den = max(float(g64.norm()), floor, 1e-30)
        rows.append((k, float((gh.double() - g64).norm()) / den, float((out["f32"][k] - g64).norm()) / den, gh.dim() == 1))
    # weight tensors: 4x / 2e-4 (4x = the per-launch pin of the split products against the fp32-MFMA path); 1-D parameters (conv bias, BatchNorm gamma / beta: column sums under cancellation, where the
    # residual of the six-product split adds coherently -- profiles/r06_masked_precision_class.txt): 8x / 5e-4
    bad = [r for r in rows if not r[1] <= (max(8.0 * r[2], 5e-4) if r[3] else max(4.0 * r[2], 2e-4))]
    ratio = np.array([r[1] / max(r[2], 1e-12) for r in rows if r[2] > 1e-7])
    worst = max(rows, key=lambda r: r[1] / max(r[2], 2e-4 / 3))
    return {"parameters": len(rows), "decision_sites": len(decisions),
            "flipped_elements_note": "decisions are the HIP path's; both oracle passes replay them",
            "bad": {r[0]: {"hip": r[1], "cpu_fp32_oracle": r[2]} for r in bad[:40]}, "n_bad": len(bad),
            "worst": {"name": worst[0], "hip": worst[1], "cpu_fp32_oracle": worst[2]},
            "ratio_gmean": float(np.exp(np.log(np.maximum(ratio, 1e-6)).mean())) if ratio.size else None,
            "ratio_p90": float(np.percentile(ratio, 90)) if ratio.size else None,
            "ratio_max": float(ratio.max()) if ratio.size else None,
            "bar": "every weight tensor: hip <= max(4 x cpu_fp32_oracle, 2e-4); every 1-D parameter: hip <= max(8 x cpu_fp32_oracle, "
                   "5e-4); relative L2 distance from the float64 oracle, all three passes on the HIP path's activation decisions "
                   "and upstream gradient",
            "ok": not bad}


# parameter gradients the parity block reports: first / last layer of each stream, one layer per kernel family
PARITY_GRADS = (
    "lidar_stream.downCntx.conv1.weight",               # first LiDAR layer (5 -> 32, 1x1, few-channel weight gradient)
    "lidar_stream.logits.weight",                       # last LiDAR layer (32 -> 20)
    "camera_stream_encoder.conv1.weight",               # 7x7 stem
    "camera_stream_decoder.conv.weight",                # camera head
    "lidar_stream.resBlock1.conv3.weight",              # 3x3 dilation 2 at full resolution
    "lidar_stream.resBlock2.conv4.weight",              # 2x2 dilation 2
    "lidar_stream.resBlock1.conv5.weight",              # 1x1 over a 3-way concat
    "lidar_stream.upBlock4.conv1.weight",               # 3x3 over the 16 + 64 concat behind PixelShuffle
    "lidar_stream.fusionblock_3.fuse_conv.0.weight",    # fusion block
    "lidar_stream.aspp.atrous_block12.weight",          # pruned-tap ASPP branch
    "camera_stream_encoder.layer2.0.conv1.weight",      # stride-2 3x3
    "camera_stream_encoder.layer4.2.conv2.weight",      # deepest camera layer
    "lidar_stream.resBlock3.bn2.weight",                # BatchNorm gamma (BN-backward fold)
    "lidar_stream.downCntx.conv2.bias",                 # conv bias in front of a BatchNorm
)


PEAK_HBM = 8000.0      # GB/s, MI355X_MICROARCH.md (6.3 TB/s is what a float4 copy reaches)


def plan_rooflines(plan, prof, model_tag):
    """(mfma roofline of the conv kernel, per-family HBM rooflines, kernel-time breakdown) from one op-by-op profile"""
    fam = {}
    for kind, family, flops, ms, _, nbytes in prof:
        a = fam.setdefault(family or kind, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += ms
        a[3] += nbytes
        # what the launch would cost at the achievable copy rate (6.3 TB/s, MI355X_MICROARCH.md) with a 4 us floor: a kernel
        # boundary in a replayed graph costs 3.6 us before the first byte moves, so a family of 1-8 MB launches cannot
        # reach a large fraction of 8 TB/s however good its kernels are -- `frac_of_launch_floor` is the kernel-quality number
        # (bn_bwd_reduce is two launches per op: the reduction and the fold of its partial rows)
        a[4] += max(nbytes / 6.3e12 * 1e3, 8e-3 if family == "bn_bwd_reduce" else 4e-3) if nbytes else 0.0
    convs = [k for k in ("conv_fwd", "conv_dgrad") if k in fam]
    mfma_ms = sum(fam[k][2] for k in convs)
    mfma_fl = sum(fam[k][1] for k in convs)
    n_launch = sum(fam[k][0] for k in convs)
    achieved = mfma_fl / (mfma_ms * 1e-3) / 1e12
    # which of those launches run their fp32 products as six bf16 MFMA products (three-way operand split, conv_fwd.hip
    # PIPE 5) and which on v_mfma_f32_32x32x2_f32: algorithmic flops by path, from the op arrays
    from pmf_amd import _lib as L
    split_fl = 0.0
    for ph in ("fwd", "bwd"):
        ops, kinds = getattr(plan, ph + "_ops", None), getattr(plan, ph + "_kinds", None)
        shift, meta = getattr(plan, ph + "_shift", 0), getattr(plan, "meta_" + ph, None)
        if not ops or not kinds or not meta:
            continue
        for i, m in meta.items():
            if m.get("family") in convs and kinds[i + shift] == L.OP_CONV and ops[i + shift].u.conv.w_s3:
                split_fl += m["flops"]
    share = split_fl / max(mfma_fl, 1.0)
    # (weight gradients: the N-split / 1x1 split kernels run the same six products; the few-channel tails run fp32 MFMA -- their
    # flops are < 3 % of the family, so the family is priced as split)
    share_all = (split_fl + (fam["conv_wgrad"][1] if "conv_wgrad" in fam else 0.0)) / max(
        mfma_fl + (fam["conv_wgrad"][1] if "conv_wgrad" in fam else 0.0), 1.0)
    # the headline fraction covers ALL matrix work of the pass: forward, input-gradient AND
    # weight-gradient launches; the per-family split rides along
    allk = [k for k in ("conv_fwd", "conv_dgrad", "conv_wgrad") if k in fam]
    # the stage-2 reductions of the weight-gradient slabs (wgrad_reduce_multi_k / wgrad_reduce_k) ARE weight-gradient time: their
    # launches count in the denominator of the headline fraction and of the weight-gradient family
    red_ms = sum(v[2] for k, v in fam.items() if k in ("OP_WGRAD_RED", "OP_WGRAD_RED_MULTI"))
    red_n = sum(v[0] for k, v in fam.items() if k in ("OP_WGRAD_RED", "OP_WGRAD_RED_MULTI"))
    if "conv_wgrad" in fam:
        fam["conv_wgrad"][2] += red_ms
        fam["conv_wgrad"][0] += red_n
        for k in ("OP_WGRAD_RED", "OP_WGRAD_RED_MULTI"):
            fam.pop(k, None)
    all_ms, all_fl, all_n = (sum(fam[k][i] for k in allk) for i in (2, 1, 0))
    achieved_all = all_fl / (all_ms * 1e-3) / 1e12
    families = {k: {"launches": fam[k][0], "algorithmic_gflop": round(fam[k][1] / 1e9, 1), "ms": round(fam[k][2], 3),
                    "achieved": round(fam[k][1] / (fam[k][2] * 1e-3) / 1e12, 2),
                    "frac": round(fam[k][1] / (fam[k][2] * 1e-3) / 1e12 / PEAK_FP32_MFMA, 4)} for k in allk}
    roof = {"bound": "mfma",
            "kernel": "the matrix kernels of the pass: conv_fwd_k (forward%s launches; fp32 arithmetic, %.0f %% of their flops as "
                      "6 bf16 MFMA products per fp32 product -- 3-way operand split, v_mfma_f32_32x32x16_bf16, fp32 accumulate "
                      "--, the rest on v_mfma_f32_32x32x2_f32)%s" % (
                          " + input-gradient" if "conv_dgrad" in fam else "", 100.0 * share,
                          " and the weight-gradient kernels (conv_wgrad_s3n_k / wgrad_1x1_s3_k on the same split products, "
                          "conv_wgrad_k on fp32 MFMA for the few-channel tails)" if "conv_wgrad" in fam else ""),
            "achieved": round(achieved_all, 3), "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s",
            "frac": round(achieved_all / PEAK_FP32_MFMA, 4),
            # the same work priced against the pipe the split products execute on: 6 bf16 MFMA products per fp32 product
            # (for the share of the flops that runs split; the fp32-MFMA share is priced at the fp32 peak)
            "bf16_pipe_frac": round(achieved_all * (6.0 * share_all / PEAK_BF16_MFMA + (1.0 - share_all) / PEAK_FP32_MFMA), 4),
            "stage2_reduce_ms_included": round(red_ms, 3),
            "families": families,
            "conv_fwd_k_only": {"achieved": round(achieved, 3), "frac": round(achieved / PEAK_FP32_MFMA, 4)},
            "peak_note": "peak = dense fp32 MFMA (the arithmetic type); achieved = algorithmic fp32 flops / launch time; "
                         "launch time = HIP events around 3 back-to-back launches of every conv op, one lane "
                         "(the kernel alone on the chip: profiles/*_bench_kernel_stats_one_lane.csv is the rocprofv3 "
                         "view of the same; in the timed 4-lane replay 2-4 kernels share the chip and each launch "
                         "stretches, profiles/*_bench_kernel_stats.csv)",
            "bf16_pipe": {"executed_tflops": round(achieved * (6.0 * share), 1), "peak": PEAK_BF16_MFMA,
                          "frac": round(achieved * 6.0 * share / PEAK_BF16_MFMA, 4),
                          "fp32_equivalent_ceiling_tflops": round(PEAK_BF16_MFMA / 6.0, 1)},
            "traffic": None, "traffic_source": None,
            "launches_per_iter": all_n, "avg_launch_us": round(1e3 * all_ms / all_n, 2),
            "algorithmic_gflop_per_iter": round(all_fl / 1e9, 1),
            "conv_fwd_k_launches_per_iter": n_launch, "conv_fwd_k_avg_launch_us": round(1e3 * mfma_ms / n_launch, 2)}
    hbm = []
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][3]):
        if v[3] > 0 and v[2] > 0:
            gbs = v[3] / (v[2] * 1e-3) / 1e9
            hbm.append({"kernel": k, "bound": "hbm", "launches": v[0], "achieved": round(gbs, 1), "peak": PEAK_HBM,
                        "unit": "GB/s", "frac": round(gbs / PEAK_HBM, 4), "algorithmic_mb_per_iter": round(v[3] / 1e6, 1),
                        "ms_per_iter": round(v[2], 4), "avg_mb_per_launch": round(v[3] / 1e6 / v[0], 2),
                        "frac_of_launch_floor": round(v[4] / v[2], 4)})
    detail = {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2], 3),
                  "tflops": round(v[1] / max(v[2], 1e-9) / 1e9, 2) if v[1] else None}
              for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])}
    if "conv_wgrad" in detail:
        detail["conv_wgrad"]["note"] = ("weight-gradient launches are sized for the side lanes they run on in the timed step -- %s "
                                        "workgroups, one per two CUs, so that the main lane's convolutions keep their LDS (step "
                                        "15.4 -> 15.1 ms against 256; DESIGN.md section 4 'Round 4') -- which makes the launch "
                                        "ALONE, what this line times, slower (256 workgroups: 4.7 ms / 100 TFLOP/s)"
                                        % os.environ.get("PMF_WGRAD_WGS", "128"))
    return roof, hbm, detail


def timed_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def graph_ms(fn, reps=50, replays=10):
    """GPU time per launch of fn: `reps` launches captured into one hipGraph on a side stream, HIP events around `replays`
    replays.  For launches that are shorter than the ~20 us the ctypes call path needs to issue one (the KNN vote): a
    host-side loop of such launches measures the issue rate, not the kernel."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(reps):
                fn()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(replays):
            gr.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n, argv):
    """--gpus N without a launcher environment: re-execute this script under torch.distributed.run, one rank per GPU.
    The children see WORLD_SIZE / RANK / LOCAL_RANK exactly as under the driver's own launch line; their stdout is ours
    (rank 0 prints the JSON line last).  Returns the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dist_probe(world, rank):
    """PMF_BENCH_DIST_PROBE=1 (tests/test_ddp_gloo.py): bring up the process group the way the bench does -- with gloo
    standing in for RCCL so that it runs without a GPU -- all-reduce one value and print what rank 0 saw.  Exercises the
    --gpus N self-spawn path on a CPU-only host; measures nothing."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"probe": "dist", "n_gpus": dist.get_world_size(), "backend": "gloo", "sum": t.item()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


_REAL_STDOUT = None


def emit(line):
    """the result line, on the stdout this process was started with (see main)"""
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: take the data-parallel code path (process group, range all-reduce) even with one rank")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "loader"],
                    help="train = the headline metric; infer = BASELINE configs[1]: eval-mode forward at bs=4 + KNN "
                         "post-processing per frame, reported as frames/s under its own metric name")
    ap.add_argument("--backbone", default="resnet34", help="camera backbone (resnet50: BASELINE configs[3] family)")
    ap.add_argument("--nclasses", type=int, default=20)
    ap.add_argument("--model", default="pmf", choices=["pmf", "epmf", "salsanext"],
                    help="pmf = the headline workload (BASELINE configs[2]); epmf = configs[4] (EPMF-R34), reported "
                         "under its own metric name, no CPU baseline")
    ap.add_argument("--resident-inputs", dest="fresh_inputs", action="store_false",
                    help="time the step on ONE resident batch (same device addresses every iteration).  Default: a batch at "
                         "a NEW device address every iteration -- what DataLoader + .cuda() hands the reference's trainer "
                         "(tasks/pmf/trainer.py:289-303); the captured graphs keep replaying through the plan's own "
                         "staging tensors.  The other mode's number is printed beside the headline either way")
    ap.add_argument("--knn-random-order", action="store_true",
                    help="--mode infer: KNN points in random order instead of sweep-file (azimuth-major) order")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity block of the timed plan")
    ap.add_argument("--parity-masked", action="store_true",
                    help="parity block: also compare EVERY parameter gradient with the HIP path's activation decisions injected "
                         "into the float64 / fp32 oracle passes (separates kinks from kernel defects)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32-ref", action="store_true", help="skip the fp32-MFMA-only reference measurement")
    ap.add_argument("--profile-out", default=None, help="write the per-launch HIP-event profile (one line per op)")
    args = ap.parse_args()

    probe = os.environ.get("PMF_BENCH_DIST_PROBE") == "1"
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher (one rank per GPU)
        if not probe and os.environ.get("PMF_BENCH_SHARE_GPU") != "1":
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                raise SystemExit("bench.py: --gpus %d asked for, %d device(s) visible" % (args.gpus, have))
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    # ONE JSON line on stdout: libraries that write to file descriptor 1 from native code (RCCL prints a five-line version
    # banner when its first communicator comes up) get stderr instead; the line itself goes to the descriptor we were given
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        sys.stdout = _REAL_STDOUT          # Python-level prints (the result line) keep the real stdout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher environment has WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if probe:
        return dist_probe(world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP hot path has no CPU fallback")
    # PMF_BENCH_SHARE_GPU=1 + PMF_BENCH_BACKEND=gloo (tests/test_gpu_boundary.py): a dry run of the multi-rank path on a
    # one-GPU box -- every rank on device 0, collectives over gloo (RCCL refuses two ranks on one device).  Not a
    # measurement: the line says so in "config".
    share_gpu = os.environ.get("PMF_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("PMF_BENCH_BACKEND", "nccl")
    if share_gpu:
        local = 0
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d has no device %d (%d visible)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
    if args.mode == "loader":
        return loader_bench(args, dev)
    if args.model == "salsanext":
        return salsanext_bench(args, dev, multi, rank, world)
    from pmf_amd.engine import TrainEngine, EPMFEngine
    from pmf_amd.models import PMFNet, EPMFNet

    torch.manual_seed(1)                 # tasks/pmf/main.py:20-21: same seed on every rank
    torch.cuda.manual_seed(1)
    net = PMFNet if args.model == "pmf" else EPMFNet
    from pmf_amd.utils.detinit import deterministic_init
    # closed-form (hash) initial weights: the same state can be given to the CPU oracle for the parity block below
    model = deterministic_init(net(5, 3, args.nclasses, 32, imagenet_pretrained=False, image_backbone=args.backbone)).to(dev)
    # pmf: fixed-weight objective (tasks/pmf/trainer.py:330-332); epmf: six terms through MultiTaskLoss(6), the sigmas in
    # the AdamW of the LiDAR stream (tasks/epmf/trainer.py:27-33,105-109,409-430, config_server_kitti.yaml use_mtloss)
    Engine = TrainEngine if args.model == "pmf" else EPMFEngine
    eng = Engine(model, args.nclasses, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                 feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=10 * 100, max_steps=49 * 100,
                 distributed=multi, device_ids=[local] if multi else None)
    feat0, mask, label = make_batch(args.bs, args.height, args.width, 1 + rank, dev, args.nclasses)   # per-rank data
    if multi and world > 1 and args.mode == "train":
        # ONE tuning pass for the job: rank 0 builds its plan first (the autotuner times every conv shape and writes its
        # choices to PMF_TUNE_CACHE), the others build theirs afterwards from that file -- every rank then runs the SAME
        # tile configurations (bit-identical arithmetic across ranks) and 7 of 8 tuning passes are saved.  Building a plan
        # issues no collective.
        from pmf_amd.models.pmf_net import _get_plan
        cache = os.environ.get("PMF_TUNE_CACHE")
        if not cache:
            cache = os.path.join("/tmp", "pmf_tune_%s_%s.txt" % (os.environ.get("MASTER_PORT", "0"), os.environ.get(
                "TORCHELASTIC_RUN_ID", "run")))
            os.environ["PMF_TUNE_CACHE"] = cache
            if rank == 0 and os.path.exists(cache):
                os.remove(cache)
        dist.barrier()
        model.train()
        if rank == 0:
            _get_plan(model, args.bs, args.height, args.width, True, dev)
            torch.cuda.synchronize()
        dist.barrier()
        if rank != 0:
            _get_plan(model, args.bs, args.height, args.width, True, dev)
    eng.time_allreduce = multi

    ring = []

    def step(fresh=None):
        if args.fresh_inputs if fresh is None else fresh:   # keep the last three batches alive: addresses rotate
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
    # warm-clock check: stream events at the start, the middle and the end of the timed region (no host sync inside it) --
    # the two halves must agree, or the clock was still ramping / throttling during the measurement
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        if i == args.steps // 2:
            ev[1].record()
        loss, _ = step()
    ev[2].record()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h1 = ev[0].elapsed_time(ev[1]) / max(args.steps // 2, 1)
    h2 = ev[1].elapsed_time(ev[2]) / max(args.steps - args.steps // 2, 1)
    halves = {"first_half_ms_per_step": round(h1, 4), "second_half_ms_per_step": round(h2, 4),
              "drift": round(h2 / h1 - 1.0, 4) if h1 > 0 else None}
    local_ms = 1e3 * dt / args.steps
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank = None
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # per-rank view of the same region: wall time per step and the exposed all-reduce (TrainEngine._finish_allreduce)
        ex = eng.exposed_allreduce_ms() if hasattr(eng, "exposed_allreduce_ms") else None
        mine = torch.tensor([local_ms, -1.0 if ex is None else ex], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        per_rank = {"ms_per_step_min": float(allr[:, 0].min()), "ms_per_step_max": float(allr[:, 0].max()),
                    "exposed_allreduce_ms_per_step": {"rank0": float(allr[0, 1]), "max": float(allr[:, 1].max()),
                                                      "mean": float(allr[:, 1].mean())},
                    "dp_segments": int(os.environ.get("PMF_DP_SEGMENTS", "4")),
                    "gradient_payload_mb": round(4.0 * eng.flat.grad.numel() / 1e6, 1) if eng.flat is not None else None,
                    "note": "exposed tail = HIP-event time on the training stream from the end of the backward plan to the "
                            "point where every gradient range is all-reduced AND its parameters are updated (range "
                            "optimiser, engine._behind_events); the earlier ranges were reduced and updated under the "
                            "running backward plan.  PMF_OWN_OPTIM=0 / PMF_DP_MODE=segments: the stream waits for the "
                            "collectives only"}
        rng = eng.range_allreduce_us() if hasattr(eng, "range_allreduce_us") else None
        if rng:
            per_rank["range_allreduce_us_rank0"] = [{"mb": round(4.0 * nfl / 1e6, 2), "us": round(us, 1),
                                                     "gb_s": round(4.0 * nfl / max(us, 1e-9) / 1e3, 1)} for nfl, us in rng]
        try:        # which collective library this is (torch's "nccl" backend IS RCCL on ROCm) and how the job was wired
            per_rank["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:     # noqa: BLE001
            per_rank["rccl_version"] = "unavailable: %s" % e
        per_rank["backend"] = dist.get_backend()
        per_rank["env"] = {k: os.environ.get(k) for k in ("NCCL_DEBUG", "NCCL_ALGO", "NCCL_PROTO", "HSA_ENABLE_IPC_MODE_LEGACY",
                                                          "PMF_DP_MODE", "RCCL_MSCCL_ENABLE") if os.environ.get(k) is not None}
        if eng.flat is not None:
            # every rank trained on its OWN data: the parameters stay bit-identical only if every gradient range really
            # went through the all-reduce before the optimiser read it (a range reduced too early or left out diverges)
            fp = eng.flat.param
            sig = torch.stack([fp.double().sum(), fp.double().abs().sum(),
                               fp.view(torch.int32).to(torch.int64).sum().double()])
            sigs = [torch.zeros_like(sig) for _ in range(world)]
            dist.all_gather(sigs, sig)
            per_rank["parameters_identical_across_ranks"] = bool(all(torch.equal(x, sigs[0]) for x in sigs))
            if not per_rank["parameters_identical_across_ranks"]:
                raise SystemExit("bench.py: the ranks' parameters diverged -- a gradient range missed its all-reduce: %r" % sigs)
    dt = t.item()
    loss_val = float(loss)
    if not np.isfinite(loss_val):
        raise SystemExit("bench.py: non-finite loss %r" % loss_val)

    # the other input mode, right behind the timed region (same process, same plan, warm clocks): all ranks take part
    ko = max(1, min(10, args.steps))
    for _ in range(2):
        step(not args.fresh_inputs)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(ko):
        step(not args.fresh_inputs)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    to = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(to, op=dist.ReduceOp.MAX)
    other_mode = {"fresh_input_addresses": not args.fresh_inputs, "value": world * ko / to.item(), "unit": "iter/s", "steps": ko}

    parity = None
    if rank == 0 and world == 1 and not args.no_parity:
        parity = parity_block(args, eng, model, feat0, mask, label)

    roof, detail, hbm = None, None, None
    if rank == 0 and not args.no_roofline:
        # one extra iteration with a HIP event pair around every launch of the forward and backward plans
        plan = next(iter(model._plans.values()))
        eng.model.train()
        # rank-local measurement: no collective may be issued here (the other ranks are not taking part)
        hooks = _detach_dp_hooks(model)
        pcd, rgb = eng.prepare(feat0.clone(), mask)
        total = eng.forward_loss(pcd, rgb, label.long())[0]
        # (each pass is profiled twice and the second reading kept: the block may follow seconds of CPU-only work --
        # parity oracle, CPU baseline -- and the first launches then run at idle clocks: one collected line showed the
        # forward launches at 62 TFLOP/s next to 94 for the input gradients profiled a moment later)
        plan.run_profiled("forward")
        prof_f = plan.run_profiled("forward")       # re-runs the forward plan op by op (same inputs)
        total.backward()                             # normal backward (stages the upstream gradients) ...
        plan.run_profiled("backward")
        prof_b = plan.run_profiled("backward")      # ... then the backward plan again, op by op
        _attach_dp_hooks(model, hooks)
        if args.profile_out:
            with open(args.profile_out, "w") as f:
                for ph, prof in (("fwd", prof_f), ("bwd", prof_b)):
                    for kind, family, flops, ms, name, _ in prof:
                        f.write("%s %-18s %-12s %-28s %9.1f us %8.2f GF %7.2f TF/s\n" % (
                            ph, kind, family or "-", name, ms * 1e3, flops / 1e9, flops / max(ms, 1e-9) / 1e9))
        roof, hbm, detail = plan_rooflines(plan, prof_f + prof_b, args.model)
        # the fused objective runs outside the plans (loss/fused.py): two probability maps read, two gradient maps
        # written, 2 x C x N x H x W sort keys written and read back permuted
        from pmf_amd.loss import pmf_total_loss_fused
        lp = torch.softmax(torch.randn(args.bs, args.nclasses, args.height, args.width, device=dev), 1).requires_grad_(True)
        cp = torch.softmax(torch.randn(args.bs, args.nclasses, args.height, args.width, device=dev), 1).requires_grad_(True)

        def loss_step():
            t, _ = pmf_total_loss_fused(lp, cp, label.long(), eng.focal.alpha, 1.0, 0.5, 0.7, eng.focal.gamma)
            t.backward()
        ms = timed_ms(loss_step)
        nb = 4.0 * lp.numel() * (2 + 2 + 2 * 3)
        hbm.append({"kernel": "fused objective (loss_pixel_k + Lovasz sort + gradient scatter, both heads, fwd+bwd)",
                    "bound": "hbm", "launches": None, "achieved": round(nb / ms / 1e6, 1), "peak": PEAK_HBM, "unit": "GB/s",
                    "frac": round(nb / ms / 1e6 / PEAK_HBM, 4), "algorithmic_mb_per_iter": round(nb / 1e6, 1),
                    "ms_per_iter": round(ms, 4), "launches": 17,
                    # (loss_pixel, 4 x (histogram, scan, scatter), unkey, Lovasz sums, Lovasz gradient, fold: a chain of 17
                    # dependent launches at 4 us each is the floor of this form; the sort is what a fifth of the time goes to)
                    "frac_of_launch_floor": round(max(nb / 6.3e12 * 1e3, 4e-3 * 17) / ms, 4)})
        # HBM-side bytes per conv launch come from a separate rocprofv3 --pmc run of this command (counters cannot be
        # read from inside the process); tools/pmc_traffic.py wrote the summary that is committed under profiles/
        headline = args.model == "pmf" and args.backbone == "resnet34" and args.nclasses == 20 and \
            (args.height, args.width, args.bs) == (64, 2048, 2)
        import glob
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        for tp in sorted(glob.glob(os.path.join(pdir, "r??_pmc_traffic.json")), reverse=True):     # the newest round first
            if headline and os.path.exists(tp):
                with open(tp) as f:
                    tj = json.load(f)
                roof["traffic"] = round(tj["hbm_bytes_per_launch"])
                roof["traffic_source"] = "profiles/%s (%s); collected at commit %s" % (
                    os.path.basename(tp), tj["method"], tj.get("commit", "unknown (before round 4)"))
                break
        # ---- the roofline of the step that was TIMED (four lanes, hipGraph replay): every algorithmic flop of the iteration
        # (forward + input-gradient + weight-gradient launches) over ms_per_step of the timed region above
        tot_fl = sum(v["gflop"] for v in detail.values() if v.get("gflop")) * 1e9
        step_ms = 1e3 * dt / args.steps
        ach = tot_fl / (step_ms * 1e-3) / 1e12
        split_share = roof["bf16_pipe"]["executed_tflops"] / max(roof["achieved"] * 6.0, 1e-9)
        roof["in_step"] = {
            "what": "all matrix work of one iteration / ms_per_step of the timed region (lanes + graph replay, everything "
                    "else of the step included in the denominator)",
            "algorithmic_gflop_per_iter": round(tot_fl / 1e9, 1), "ms_per_step": round(step_ms, 4),
            "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA, 4),
            "bf16_pipe_frac": round(ach * 6.0 * split_share / PEAK_BF16_MFMA, 4),
            "non_matrix_kernel_ms_one_lane": round(sum(v["ms"] for k, v in detail.items() if not v.get("gflop")), 3)}
        ip = (sorted(glob.glob(os.path.join(pdir, "r??_in_step.json")), reverse=True) or [""])[0]
        if headline and os.path.exists(ip):
            with open(ip) as f:
                roof["in_step"]["trace"] = json.load(f)      # rocprofv3 view of the four-lane replay (tools/in_step.py)

    # the same training step with every convolution on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32; PMF_CONV_F32=1): the
    # split-bf16 products carry fp32-class error (tests), this line shows what they buy
    f32_only = None
    if rank == 0 and world == 1 and args.mode == "train" and not args.no_f32_ref:
        os.environ["PMF_CONV_F32"] = "1"
        try:
            torch.manual_seed(1)
            model2 = net(5, 3, args.nclasses, 32, imagenet_pretrained=False, image_backbone=args.backbone).to(dev)
            eng2 = Engine(model2, args.nclasses, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                          feature_mean=KITTI_MEAN, feature_std=KITTI_STD, warmup_steps=10 * 100, max_steps=49 * 100)
            for _ in range(6):
                eng2.train_step(feat0.clone(), mask, label)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(15):
                eng2.train_step(feat0.clone(), mask, label)
            torch.cuda.synchronize()
            f32_only = {"value": 15 / (time.perf_counter() - t1), "unit": "iter/s", "steps": 15,
                        "note": "PMF_CONV_F32=1: all conv / weight-gradient products on v_mfma_f32_32x32x2_f32"}
            del eng2, model2
        finally:
            del os.environ["PMF_CONV_F32"]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.bs, args.height, args.width, args.model, args.backbone, args.nclasses)

    if rank == 0:
        iters = world * args.steps
        tag = ("PMF" if args.model == "pmf" else "EPMF")
        bb = {"resnet34": "ResNet34", "resnet50": "ResNet50"}.get(args.backbone, args.backbone)
        out = {
            "metric": "train iters/sec %s-%s %dx%d bs=%d/GPU (full iteration: fwd + %s + bwd + "
                      "AdamW/SGD steps)" % (tag, bb, args.height, args.width, args.bs,
                                            "5-term loss" if args.model == "pmf" else "6-term multi-task loss"),
            "value": iters / dt, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": ARITH_NOTE,
            "config": {"workload": "%s-%s, full train loop, both streams %dx%d (BASELINE configs[%d]%s), bs=%d/GPU, "
                                   "%d classes, dropout on, local-stat BN"
                                   % (tag, bb, args.height, args.width,
                                      4 if args.model == "epmf" else (3 if args.backbone == "resnet50" else 2),
                                      ", S_A" if (args.height, args.width) == (64, 2048) else "", args.bs,
                                      args.nclasses),
                       "global_batch": world * args.bs, "parallelism": "dp%d" % world,
                       "rccl_ranks": (dist.get_world_size() if multi and backend == "nccl" else 0),
                       **({"dry_run": "PMF_BENCH_SHARE_GPU=1: %d ranks share GPU 0, collectives over %s -- a functional "
                                      "check of the multi-rank path, NOT a measurement" % (world, backend)}
                          if share_gpu or backend != "nccl" else {}),
                       "samples_per_s": world * args.bs * args.steps / dt, "final_loss": loss_val,
                       "fresh_input_addresses": bool(args.fresh_inputs),
                       "init": "closed-form hash weights (pmf_amd.utils.detinit), %d training iterations before the parity block" % (args.warmup + args.steps + ko + 2),
                       "graphs_captured": len(next(iter(model._plans.values()))._graphs)},
            "warm_clock_check": halves, "data_parallel": per_rank,
            "parity": parity, "other_input_mode": other_mode,
            "roofline": roof, "roofline_hbm": hbm, "cpu_baseline": cpu, "fp32_mfma_only": f32_only,
            "kernel_time_breakdown": detail,
        }
        try:       # RCCL prints its version banner through C stdio (buffered when piped): push it out first so that the
            import ctypes           # JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        emit(json.dumps(out))
    if multi:
        dist.barrier()             # rank 0 measured its roofline alone: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
