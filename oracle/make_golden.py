#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES (CPU) in the build container.

TEST INFRASTRUCTURE.  Run from the repo root:  python oracle/make_golden.py
Needs /root/reference (read-only).  The fixtures hold inputs-by-recipe (closed-form
``pmf_amd.utils.detinit``) and expected OUTPUTS only -- no reference source text.

How the reference is imported (it cannot be imported as a package: tensorboardX,
torchvision, nuscenes-devkit, cv2 are absent -- SURVEY.md 8c):
  * pc_processor/models/salsanext.py, postproc/knn.py, loss/*.py, metrics/iou_eval.py,
    utils/warmup_lr.py, dataset/semantic_kitti/parser.py: loaded by file path.
  * pc_processor/models/pmf_net.py needs ``torchvision.models.resnet``; torchvision is a
    THIRD-PARTY dependency that is not installed, so the camera backbone is supplied by this
    repo's restatement (oracle/pmf_torch.py: BasicBlock/Bottleneck).  Everything else in
    PMFNet (fusion blocks, ASPP, decoder, SalsaNext trunk, stem replacement) is reference code.
    => camera-backbone parity is UNPINNED (stated in DESIGN.md); the rest is pinned.
  * dataset/perspective_view_loader.py needs ``torchvision.transforms`` only to CONSTRUCT
    transforms that the return_uproj path never applies; name-only placeholders are registered.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from pmf_amd.utils.detinit import deterministic_init, det_tensor, synthetic_batch  # noqa: E402
from oracle import pmf_torch as O  # noqa: E402
from oracle import knn_ref, loader_ref  # noqa: E402,F401
from oracle.cases import knn_case, RANGE_CASES  # noqa: E402


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def import_reference():
    # synthetic parent packages so relative imports resolve
    for name in ("refpc", "refpc.models"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")

    def _mk(name):
        def ctor(pretrained=False):
            blk, cnt = O.RESNET_CFG[name]
            net = O.ResNet(3, name)
            net.relu = torch.nn.ReLU(inplace=True)
            net.maxpool = torch.nn.MaxPool2d(3, 2, 1)
            return net
        return ctor
    for n in O.RESNET_CFG:
        setattr(tvr, n, _mk(n))
    tvt = types.ModuleType("torchvision.transforms")
    for n in ("ColorJitter", "Pad", "Compose", "RandomHorizontalFlip", "RandomRotation",
              "RandomCrop", "CenterCrop"):
        setattr(tvt, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    tv.models, tv.transforms, tvm.resnet = tvm, tvt, tvr
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm,
                        "torchvision.models.resnet": tvr, "torchvision.transforms": tvt})
    R = types.SimpleNamespace()
    R.salsanext = _load("refpc.models.salsanext", "pc_processor/models/salsanext.py")
    R.pmf_net = _load("refpc.models.pmf_net", "pc_processor/models/pmf_net.py")
    R.knn = _load("refpc_knn", "pc_processor/postproc/knn.py")
    R.focal = _load("refpc_focal", "pc_processor/loss/focal_softmax.py")
    R.lovasz = _load("refpc_lovasz", "pc_processor/loss/lovasz_softmax.py")
    R.iou = _load("refpc_iou", "pc_processor/metrics/iou_eval.py")
    R.warmup = _load("refpc_warmup", "pc_processor/utils/warmup_lr.py")
    R.parser = _load("refpc_parser", "pc_processor/dataset/semantic_kitti/parser.py")
    # loader: needs `pc_processor.dataset.preprocess.augmentor`
    for name in ("pc_processor", "pc_processor.dataset", "pc_processor.dataset.preprocess"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    aug = _load("pc_processor.dataset.preprocess.augmentor", "pc_processor/dataset/preprocess/augmentor.py")
    sys.modules["pc_processor.dataset.preprocess"].augmentor = aug
    R.loader = _load("refpc_loader", "pc_processor/dataset/perspective_view_loader.py")
    return R


def set_p0(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
        if isinstance(m, O.DropSite):
            m.p = 0.0


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def grad_digest(model):
    """per-parameter L2 norm + 8 sampled entries (fixed stride)."""
    out = {}
    for k, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        idx = torch.linspace(0, g.numel() - 1, 8).long()
        out[k] = np.concatenate([[g.double().norm().item()], g[idx].double().numpy()])
    return out


def blocks(R):
    """G1/G2: per-block eval + train(p=0) outputs, running stats, input/param grads."""
    S, Pn = R.salsanext, R.pmf_net
    cases = {
        "ResContextBlock": (lambda: S.ResContextBlock(8, 32), [(2, 8, 16, 32)]),
        "ResBlock_pool": (lambda: S.ResBlock(32, 64, 0.2, pooling=True, drop_out=False), [(2, 32, 16, 32)]),
        "ResBlock_nopool": (lambda: S.ResBlock(64, 64, 0.2, pooling=False), [(2, 64, 8, 16)]),
        "UpBlock": (lambda: S.UpBlock(64, 32, 0.2), [(2, 64, 8, 16), (2, 64, 16, 32)]),
        "Fusion": (lambda: Pn.ResidualBasedFusionBlock(32, 64), [(2, 32, 8, 16), (2, 64, 8, 16)]),
        "ASPP": (lambda: Pn.ASPP(64, 64), [(2, 64, 8, 40)]),
        "RGBDecoder": (lambda: Pn.RGBDecoder([16, 32, 64, 128], 20, 16),
                       [[(1, 16, 16, 32), (1, 32, 8, 16), (1, 64, 4, 8), (1, 128, 2, 4)]]),
    }
    out = {}
    for name, (ctor, shapes) in cases.items():
        torch.manual_seed(0)
        m = deterministic_init(ctor())
        set_p0(m)

        def mk_inputs():
            ins = []
            for i, s in enumerate(shapes):
                if isinstance(s, list):
                    ins.append([det_tensor("%s.in%d.%d" % (name, i, j), ss).requires_grad_(True)
                                for j, ss in enumerate(s)])
                else:
                    ins.append(det_tensor("%s.in%d" % (name, i), s).requires_grad_(True))
            return ins
        for mode in ("eval", "train"):
            m.train(mode == "train")
            ins = mk_inputs()
            y = m(*ins)
            ys = list(y) if isinstance(y, (tuple, list)) else [y]
            for j, yy in enumerate(ys):
                out["%s.%s.out%d" % (name, mode, j)] = f32(yy)
            if mode == "train":
                loss = sum((yy * det_tensor("%s.gout%d" % (name, j), yy.shape)).sum() for j, yy in enumerate(ys))
                loss.backward()
                flat = [t for i in ins for t in (i if isinstance(i, list) else [i])]
                for j, t in enumerate(flat):
                    out["%s.train.gin%d" % (name, j)] = f32(t.grad)
                for k, p in m.named_parameters():
                    out["%s.train.gparam.%s" % (name, k)] = f32(p.grad)
                for k, b in m.named_buffers():
                    if "running" in k:
                        out["%s.train.buf.%s" % (name, k)] = f32(b)
    np.savez_compressed(os.path.join(OUT, "g12_blocks.npz"), **out)
    print("g12_blocks: %d arrays" % len(out))


def whole_net(R):
    """G3: PMFNet-R34 (20 cls) / R50 (17 cls) / SalsaNext at 32x64; config-1 (64x512, bs 1) losses."""
    out = {}
    alpha = torch.linspace(0.2, 1.0, 20)
    alpha[0] = 0
    for tag, kw, n, h, w in (("r34", dict(nclasses=20, image_backbone="resnet34"), 2, 32, 64),
                             ("r50", dict(nclasses=17, image_backbone="resnet50"), 1, 32, 64)):
        m = deterministic_init(R.pmf_net.PMFNet(pcd_channels=5, img_channels=3, base_channels=32,
                                                imagenet_pretrained=False, **kw))
        set_p0(m)
        hooks = {}
        m.lidar_stream.logits.register_forward_hook(lambda mod, i, o: hooks.__setitem__("lidar", o))
        m.camera_stream_decoder.conv.register_forward_hook(lambda mod, i, o: hooks.__setitem__("cam", o))
        pcd, rgb, label, _ = synthetic_batch(n, h, w, kw["nclasses"], seed=1)
        m.eval()
        with torch.no_grad():
            lp, cp = m(pcd, rgb)
        out["%s.eval.lidar_logits" % tag] = f32(hooks["lidar"])
        out["%s.eval.cam_logits" % tag] = f32(hooks["cam"])
        out["%s.eval.lidar_prob" % tag] = f32(lp)
        out["%s.eval.cam_prob" % tag] = f32(cp)
        out["%s.nparams" % tag] = np.array([sum(p.numel() for p in m.parameters())])
        out["%s.keys" % tag] = np.array(sorted(m.state_dict().keys()))
        if tag != "r34":
            continue
        m.train()
        lp, cp = m(pcd, rgb)
        out["r34.train.lidar_logits"] = f32(hooks["lidar"])
        out["r34.train.cam_logits"] = f32(hooks["cam"])
        lov = R.lovasz.Lovasz_softmax(ignore=0)
        foc = R.focal.FocalSoftmaxLoss(20, gamma=2, alpha=alpha.numpy(), softmax=False)
        total, terms = reference_loss(lov, foc, lp, cp, label)
        total.backward()
        out["r34.train.losses"] = np.array([total.item()] + [terms[k].item() for k in
                                                             ("foc", "lov", "foc_cam", "lov_cam", "per")])
        for k, v in grad_digest(m).items():
            out["r34.train.gdig." + k] = v
        for k, b in m.named_buffers():
            if k.endswith("running_mean") and ("downCntx.bn1" in k or "layer4.2.bn2" in k or "up_1a" in k):
                out["r34.train.buf." + k] = f32(b)
    # SalsaNext stand-alone (API row b)
    s = deterministic_init(R.salsanext.SalsaNext(in_channels=5, nclasses=20, base_channels=32))
    s.eval()
    pcd, _, _, _ = synthetic_batch(1, 32, 64, 20, seed=2)
    with torch.no_grad():
        out["salsanext.eval.prob"] = f32(s(pcd))
    np.savez_compressed(os.path.join(OUT, "g3_wholenet.npz"), **out)
    print("g3_wholenet: %d arrays" % len(out))


def reference_loss(lov, foc, lp, cp, label, lambda_=1.0, gamma_=0.5, tau=0.7):
    """tasks/pmf/trainer.py:303-332 driven with the reference's loss modules."""
    import math
    kl = torch.nn.KLDivLoss(reduction="none")
    mask = label.gt(0)
    ncls = lp.shape[1]
    llog = torch.log(lp.clamp(min=1e-8))
    pent = -(lp * llog).sum(1) / math.log(ncls)
    clog = torch.log(cp.clamp(min=1e-8))
    ient = -(cp * clog).sum(1) / math.log(ncls)
    t = {"foc": foc(lp, label, mask=mask), "lov": lov(lp, label),
         "foc_cam": foc(cp, label, mask=mask), "lov_cam": lov(cp, label)}
    pc, ic = 1 - pent, 1 - ient
    d = pc - ic
    wp = d.gt(0).float() * d.abs() * pc.ge(tau).float()
    wi = d.lt(0).float() * d.abs() * ic.ge(tau).float()
    t["per"] = (kl(llog, cp) * wi.unsqueeze(1)).mean() + (kl(clog, lp) * wp.unsqueeze(1)).mean()
    total = t["foc"] + t["lov"] * lambda_ + t["foc_cam"] + t["lov_cam"] * lambda_ + t["per"] * gamma_
    return total, t


def losses_metrics(R):
    """G6: focal / lovasz values+grads, IOUEval stats, WarmupCosineLR trace."""
    out = {}
    n, c, h, w = 2, 20, 16, 32
    logits = det_tensor("g6.logits", (n, c, h, w), -3, 3)
    logits2 = det_tensor("g6.logits2", (n, c, h, w), -3, 3)
    _, _, label, _ = synthetic_batch(n, h, w, c, seed=3, fill=0.4)
    alpha = torch.linspace(0.2, 1.0, c)
    alpha[0] = 0
    lov = R.lovasz.Lovasz_softmax(ignore=0)
    foc = R.focal.FocalSoftmaxLoss(c, gamma=2, alpha=alpha.numpy(), softmax=False)
    a = logits.clone().requires_grad_(True)
    b = logits2.clone().requires_grad_(True)
    lp, cp = torch.softmax(a, 1), torch.softmax(b, 1)
    total, terms = reference_loss(lov, foc, lp, cp, label)
    total.backward()
    out["loss.values"] = np.array([total.item()] + [terms[k].item() for k in ("foc", "lov", "foc_cam", "lov_cam", "per")])
    out["loss.grad_a"] = f32(a.grad)
    out["loss.grad_b"] = f32(b.grad)
    ev = R.iou.IOUEval(c, torch.device("cpu"), ignore=[0], is_distributed=False)
    ev.addBatch(lp.argmax(1), label)
    ev.addBatch(cp.argmax(1), label)
    out["iou.conf"] = ev.conf_matrix.numpy()
    for nm, fn in (("iou", ev.getIoU), ("acc", ev.getAcc), ("recall", ev.getRecall)):
        mean, per = fn()
        out["iou.%s" % nm] = np.concatenate([[mean.item()], per.numpy()])
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.001, momentum=0.9)
    sch = R.warmup.WarmupCosineLR(opt, lr=0.001, warmup_steps=5, momentum=0.9, max_steps=10)
    trace = []
    for _ in range(15):
        opt.step()
        sch.step()
        trace.append(opt.param_groups[0]["lr"])
    out["lr.trace"] = np.array(trace)
    np.savez_compressed(os.path.join(OUT, "g6_losses.npz"), **out)
    print("g6_losses: %d arrays" % len(out))


def knn(R):
    """G4: reference KNN labels on seeded synthetic cases (inputs regenerated by recipe in tests)."""
    out = {}
    for tag, seed, h, w, npts, kw in (("a", 11, 64, 512, 20000, {}), ("b", 12, 48, 160, 6000, {}),
                                      ("ties", 13, 32, 64, 3000, {"quantize": True})):
        pr, ur, am, px, py = knn_case(seed, h, w, npts, **kw)
        post = R.knn.KNN({"knn": 5, "search": 5, "sigma": 1.0, "cutoff": 1.0}, 20)
        lab = post(torch.from_numpy(pr), torch.from_numpy(ur), torch.from_numpy(am),
                   torch.from_numpy(px), torch.from_numpy(py))
        out["knn.%s.labels" % tag] = lab.numpy().astype(np.int16)
    w = (1 - R.knn.get_gaussian_kernel(5, 1.0, 1)).numpy()
    out["knn.inv_gauss"] = w.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g4_knn.npz"), **out)
    print("g4_knn: %d arrays" % len(out))


def loader(R):
    """G5: reference PerspectiveViewLoader (return_uproj path) + SemanticKitti.mapLidar2Camera."""
    out = {}
    from PIL import Image
    for tag, seed, npts, h, w in (("a", 0, 5000, 96, 320), ("b", 5, 20000, 64, 208)):
        M, pts, sem, img, lut = loader_ref.synthetic_frame(seed, npts, h, w)
        ds = object.__new__(R.parser.SemanticKitti)
        ds.has_image = True
        ds.proj_matrix = {"00": M}
        ds.class_map_lut = lut
        ds.loadDataByIndex = lambda i: (pts, sem, np.zeros_like(sem))
        ds.loadImage = lambda i: Image.fromarray(img)
        ds.parsePathInfoByIndex = lambda i: ("00", "000000")
        ds.pointcloud_files = [None]
        cfg = {"augmentation": {}, "sensor": {"proj_h": h, "proj_w": w, "proj_ht": h, "proj_wt": w,
                                              "h_pad": 0, "w_pad": 0}}
        ld = R.loader.PerspectiveViewLoader(ds, cfg, is_train=False, return_uproj=True)
        feat, mask, label, xd, yd, depth = ld[0]
        out["loader.%s.proj" % tag] = np.concatenate([feat.numpy(), mask.numpy()[None], label.numpy()[None]], 0)
        out["loader.%s.x_data" % tag] = xd.numpy()
        out["loader.%s.y_data" % tag] = yd.numpy()
        out["loader.%s.depth" % tag] = depth.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_loader.npz"), **out)
    print("g5_loader: %d arrays" % len(out))


def epmf(R):
    """G8: the reference's EPMFNet (epmf_net.py) -- SparseVariantConv + stride-2 ResContextBlock at block level, whole
    net eval/train (dropout p=0) at 64x128: logits, probabilities, gradient digest, running stats; MultiTaskLoss."""
    out = {}
    E = _load("refpc.models.epmf_net", "pc_processor/models/epmf_net.py")
    mt = _load("refpc_mtloss", "pc_processor/loss/multi_task_loss.py")
    # block level: ResContextBlock(8->32, stride 2) on a sparse input, train mode, input + parameter gradients
    blk = deterministic_init(E.ResContextBlock(8, 32, stride=2))
    blk.train()
    pcd, _, _, mask = synthetic_batch(2, 16, 32, 20, seed=4, fill=0.3)
    x = torch.cat((pcd, torch.zeros(2, 3, 16, 32)), 1).requires_grad_(True)
    y = blk(x)
    gy = det_tensor("g8.blk.gy", tuple(y.shape))
    (y * gy).sum().backward()
    out["blk.out"] = f32(y)
    out["blk.gin"] = f32(x.grad)
    for k, v in grad_digest(blk).items():
        out["blk.gdig." + k] = v
    out["blk.bn1.running_mean"] = f32(blk.bn1.running_mean)
    out["blk.bn2.running_var"] = f32(blk.bn2.running_var)
    # whole net
    m = deterministic_init(E.EPMFNet(pcd_channels=5, img_channels=3, nclasses=20, base_channels=32,
                                     imagenet_pretrained=False, image_backbone="resnet34"))
    set_p0(m)
    hooks = {}
    m.lidar_stream.logits.register_forward_hook(lambda mod, i, o: hooks.__setitem__("lidar", o))
    m.camera_stream_decoder.conv.register_forward_hook(lambda mod, i, o: hooks.__setitem__("cam", o))
    pcd, rgb, label, _ = synthetic_batch(2, 64, 128, 20, seed=1, fill=0.3)
    m.eval()
    with torch.no_grad():
        lp, cp = m(pcd, rgb)
    out["eval.lidar_logits"], out["eval.cam_logits"] = f32(hooks["lidar"]), f32(hooks["cam"])
    out["nparams"] = np.array([sum(p.numel() for p in m.parameters())])
    out["keys"] = np.array(sorted(m.state_dict().keys()))
    m.train()
    lp, cp = m(pcd, rgb)
    out["train.lidar_logits"] = f32(hooks["lidar"])
    alpha = torch.linspace(0.2, 1.0, 20)
    alpha[0] = 0
    lov = R.lovasz.Lovasz_softmax(ignore=0)
    foc = R.focal.FocalSoftmaxLoss(20, gamma=2, alpha=alpha.numpy(), softmax=False)
    total, terms = reference_loss(lov, foc, lp, cp, label)
    total.backward()
    out["train.losses"] = np.array([total.item()] + [terms[k].item() for k in ("foc", "lov", "foc_cam", "lov_cam", "per")])
    for k, v in grad_digest(m).items():
        out["train.gdig." + k] = v
    for k, b in m.named_buffers():
        if k.endswith("running_mean") and ("downCntx3.bn2" in k or "extraUpSample" in k):
            out["train.buf." + k] = f32(b)
    # MultiTaskLoss (multi_task_loss.py:14-19)
    mtl = mt.MultiTaskLoss(6)
    ls = [torch.tensor(v, requires_grad=True) for v in (0.7, 1.3, 0.2, 2.1, 0.9, 0.05)]
    tot = mtl(ls)
    tot.backward()
    out["mtl.total"] = np.array([tot.item()])
    out["mtl.gsigma"] = f32(mtl.sigma.grad)
    out["mtl.gloss"] = np.array([l.grad.item() for l in ls])
    np.savez_compressed(os.path.join(OUT, "g8_epmf.npz"), **out)
    print("g8_epmf: %d arrays" % len(out))


def loader_v2(R):
    """G9: reference PerspectiveViewLoaderV2 (return_uproj path) + SemanticKitti.mapLidar2CameraCropYaw."""
    from PIL import Image
    cv2 = types.ModuleType("cv2")
    cv2.rotate = lambda *a, **k: None          # imported by name only (perspective_view_loader_v2.py:2), never called
    sys.modules["cv2"] = cv2
    V2 = _load("refpc_loader_v2", "pc_processor/dataset/perspective_view_loader_v2.py")
    out = {}
    for tag, seed, npts, h, w in (("a", 0, 5000, 96, 320), ("b", 5, 20000, 64, 208)):
        M, pts, sem, img, lut = loader_ref.synthetic_frame(seed, npts, h, w)
        ds = object.__new__(R.parser.SemanticKitti)
        ds.has_image = True
        ds.proj_matrix = {"00": M}
        ds.class_map_lut = lut
        ds.fov_left, ds.fov_right = -45 / 180.0 * np.pi, 45 / 180.0 * np.pi
        ds.loadDataByIndex = lambda i: (pts, sem, np.zeros_like(sem))
        ds.loadImage = lambda i: Image.fromarray(img)
        ds.parsePathInfoByIndex = lambda i: ("00", "000000")
        ds.pointcloud_files = [None]
        cfg = {"PVconfig": {"proj_h": h, "proj_w": w, "proj_ht": h, "proj_wt": w, "img_jitter": [0.4, 0.4, 0.4]}}
        ld = V2.PerspectiveViewLoaderV2(ds, cfg, is_train=False, return_uproj=True)
        proj, xy, depth, keep, pc = ld[0]
        out["v2.%s.proj" % tag] = proj.numpy()
        out["v2.%s.xy" % tag] = xy.numpy()
        out["v2.%s.depth" % tag] = depth.numpy()
        out["v2.%s.keep" % tag] = keep.numpy()
    np.savez_compressed(os.path.join(OUT, "g9_loader_v2.npz"), **out)
    print("g9_loader_v2: %d arrays, frames %s" % (len(out), [out["v2.%s.proj" % t].shape for t in ("a", "b")]))


def range_loader(R):
    """G10: reference RangeProjection / SalsaNextLoader (return_uproj path) and Augmentor on synthetic sweeps."""
    import random
    from oracle.cases import lidar_sweep
    proj = _load("pc_processor.dataset.preprocess.projection", "pc_processor/dataset/preprocess/projection.py")
    sys.modules["pc_processor.dataset.preprocess"].projection = proj
    SL = _load("refpc_salsanext_loader", "pc_processor/dataset/salsanext_loader.py")
    aug = sys.modules["pc_processor.dataset.preprocess.augmentor"]
    out = {}
    for tag, seed, npts, cfg in RANGE_CASES:
        pts, sem, lut = lidar_sweep(seed, npts, cfg["sensor"]["fov_up"], cfg["sensor"]["fov_down"])
        ds = types.SimpleNamespace()
        ds.loadDataByIndex = lambda i: (pts.copy(), sem, np.zeros_like(sem))
        ds.labelMapping = lambda l: lut[l]
        ds.__len__ = lambda: 1
        ld = SL.SalsaNextLoader(ds, cfg, is_train=False, return_uproj=True)
        feat, label, mask, rng_img, ux, uy, ud = ld[0]
        pc, pr, pidx, pmask = ld.projection.doProjection(pts.copy())
        out.update({"%s.feature" % tag: feat.numpy(), "%s.label" % tag: label.numpy(), "%s.mask" % tag: mask.numpy(),
                    "%s.range" % tag: rng_img.numpy(), "%s.ux" % tag: ux.numpy().astype(np.int32),
                    "%s.uy" % tag: uy.numpy().astype(np.int32), "%s.ud" % tag: ud.numpy(),
                    "%s.proj_idx" % tag: pidx, "%s.proj_mask" % tag: pmask})
        assert np.array_equal(pc[pidx >= 0], pts[pidx[pidx >= 0]]) and (pc[pidx < 0] == -1).all()
        # training path: the augmentation driven by Python's `random`
        random.seed(100 + seed)
        params = aug.AugmentParams()
        a = cfg["augmentation"]
        params.setFlipProb(p_flipx=a["p_flipx"], p_flipy=a["p_flipy"])
        params.setTranslationParams(**{k: a[k] for k in a if "trans" in k})
        params.setRotationParams(**{k: a[k] for k in a if "rot" in k})
        out["%s.augmented" % tag] = aug.Augmentor(params).doAugmentation(pts.copy())
        random.seed(100 + seed)
        ldt = SL.SalsaNextLoader(ds, cfg, is_train=True, return_uproj=False)
        ft, lt, mt = ldt[0]
        out.update({"%s.train.feature" % tag: ft.numpy(), "%s.train.label" % tag: lt.numpy(),
                    "%s.train.mask" % tag: mt.numpy()})
    np.savez_compressed(os.path.join(OUT, "g10_range.npz"), **out)
    print("g10_range: %d arrays" % len(out), {k: v.shape for k, v in out.items() if k.endswith("feature")})


def kitti_formats(R):
    """G11: the reference SemanticKitti parser on a synthetic on-disk tree (oracle/cases.py kitti_tree)."""
    import tempfile
    from oracle.cases import kitti_tree
    root = tempfile.mkdtemp()
    cfg, data = kitti_tree(root)
    ds = R.parser.SemanticKitti(root, [8, 0], cfg)
    out = {"class_map_lut": ds.class_map_lut, "class_map_lut_inv": ds.class_map_lut_inv, "cls_freq": ds.cls_freq,
           "sem_color_lut": ds.sem_color_lut, "sem_color_lut_inv": ds.sem_color_lut_inv,
           "order": np.array([os.path.relpath(f, root) for f in ds.pointcloud_files]),
           "label_order": np.array([os.path.relpath(f, root) for f in ds.label_files]),
           "image_order": np.array([os.path.relpath(f, root) for f in ds.image_files]),
           "path_info": np.array([ds.parsePathInfoByIndex(i) for i in range(len(ds.pointcloud_files))])}
    for seq, m in ds.proj_matrix.items():
        out["proj." + seq] = m
    pc, sem, inst = ds.loadDataByIndex(4)
    out.update({"f4.points": pc, "f4.sem": sem, "f4.inst": inst, "f4.mapped": ds.labelMapping(sem),
                "f4.image": np.asarray(ds.loadImage(4))})
    np.savez_compressed(os.path.join(OUT, "g11_kitti_formats.npz"), **out)
    print("g11_kitti_formats: %d arrays" % len(out))


def merge(R):
    """G12: the reference's getMergePred (tasks/pmf_eval_nuscenes/infer.py:18-38), executed from its source file.
    The module imports the nuScenes devkit at the top, so only that function is compiled (ast) -- with Tensor.cuda
    mapped to the identity, since there is no GPU in the container where fixtures are generated."""
    import ast
    from oracle.cases import merge_case
    path = os.path.join(REF, "tasks/pmf_eval_nuscenes/infer.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "getMergePred"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        for tag, seed, pc in (("a", 0, 3000), ("b", 4, 347)):
            idx, conf, lab = merge_case(seed, pc)
            t = lambda xs: [torch.from_numpy(x) for x in xs]
            out["merge.%s" % tag] = ns["getMergePred"](t(idx), t(conf), t(lab), pc).numpy()
    finally:
        torch.Tensor.cuda = keep
    np.savez_compressed(os.path.join(OUT, "g12_merge.npz"), **out)
    print("g12_merge:", {k: (v.shape, int((v < 0).sum())) for k, v in out.items()})


def trainer_trace(R):
    """G7: two consecutive optimisation steps (AdamW lidar / SGD-Nesterov camera, trainer.py:80-98,214-219)
    on config-1 shapes (64x512, bs 1), dropout p=0."""
    out = {}
    m = deterministic_init(R.pmf_net.PMFNet(5, 3, 20, 32, False, "resnet34"))
    set_p0(m)
    m.train()
    adam = torch.optim.AdamW([{"params": m.lidar_stream.parameters()}], lr=0.001)
    sgd = torch.optim.SGD([{"params": m.camera_stream_encoder.parameters()},
                           {"params": m.camera_stream_decoder.parameters()}],
                          lr=0.001, nesterov=True, momentum=0.9, weight_decay=1e-5)
    alpha = torch.linspace(0.2, 1.0, 20)
    alpha[0] = 0
    lov = R.lovasz.Lovasz_softmax(ignore=0)
    foc = R.focal.FocalSoftmaxLoss(20, gamma=2, alpha=alpha.numpy(), softmax=False)
    pcd, rgb, label, _ = synthetic_batch(1, 64, 512, 20, seed=1)
    vals = []
    keys = ["lidar_stream.downCntx.conv1.weight", "lidar_stream.resBlock3.conv3.weight",
            "lidar_stream.logits.bias", "camera_stream_encoder.conv1.weight",
            "camera_stream_encoder.layer3.2.conv1.weight", "camera_stream_decoder.conv.weight",
            "lidar_stream.fusionblock_2.attention.4.weight"]
    for step in range(2):
        lp, cp = m(pcd, rgb)
        total, terms = reference_loss(lov, foc, lp, cp, label)
        adam.zero_grad()
        sgd.zero_grad()
        total.backward()
        adam.step()
        sgd.step()
        vals.append([total.item()] + [terms[k].item() for k in ("foc", "lov", "foc_cam", "lov_cam", "per")])
        if step == 0:      # after ONE step (Adam's first update is lr * sign(g): well conditioned) ...
            sd = m.state_dict()
            for k in keys:
                out["trace.param1." + k] = np.array([sd[k].double().sum().item(), sd[k].double().abs().sum().item()])
            for k in ("lidar_stream.downCntx.bn1.running_mean", "lidar_stream.downCntx.bn1.running_var",
                      "camera_stream_encoder.bn1.running_mean", "camera_stream_encoder.layer2.0.bn1.running_var"):
                out["trace.buf1." + k] = sd[k].double().numpy().copy()
    out["trace.losses"] = np.array(vals)
    sd = m.state_dict()
    for k in keys:             # ... and after two
        out["trace.param." + k] = np.array([sd[k].double().sum().item(), sd[k].double().abs().sum().item()])
    np.savez_compressed(os.path.join(OUT, "g7_trace.npz"), **out)
    print("g7_trace: %d arrays" % len(out))


def _torchvision_transform_standins(tvt):
    """functional stand-ins for the torchvision.transforms classes the training path of the perspective loaders APPLIES
    (torchvision is absent).  ColorJitter: torchvision's published get_params draw order, pixel operations by Pillow
    itself (ImageEnhance / HSV convert -- what torchvision's PIL backend calls).  Flip / rotate / crop / pad / compose:
    oracle/tensor_aug_ref.py (torch's own grid_sample).  => the fixture pins the reference LOADER code + Pillow; the
    torchvision layer is restated (unpinned)."""
    from PIL import Image, ImageEnhance
    from oracle import color_jitter_ref as CJ
    from oracle import tensor_aug_ref as TA

    class ColorJitter:
        def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
            self.ranges = CJ.jitter_ranges(brightness, contrast, saturation, hue)

        def __call__(self, img):
            order, fac = CJ.draw_params(self.ranges)
            for fn_id in order:
                f = fac[fn_id]
                if f is None:
                    continue
                if fn_id == 0:
                    img = ImageEnhance.Brightness(img).enhance(f)
                elif fn_id == 1:
                    img = ImageEnhance.Contrast(img).enhance(f)
                elif fn_id == 2:
                    img = ImageEnhance.Color(img).enhance(f)
                else:
                    h, s_, v = img.convert("HSV").split()
                    np_h = np.array(h, dtype=np.uint8)
                    with np.errstate(over="ignore"):
                        np_h += np.uint8(int(f * 255) & 0xff)
                    img = Image.merge("HSV", (Image.fromarray(np_h, "L"), s_, v)).convert("RGB")
            return img

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class RandomHorizontalFlip:
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, x):
            return x.flip(-1) if torch.rand(1) < self.p else x

    class RandomRotation:
        def __init__(self, degrees):
            self.d = float(degrees)

        def __call__(self, x):
            angle = float(torch.empty(1).uniform_(-self.d, self.d).item())
            return TA.rotate_nearest(x, angle)

    class RandomCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            h, w = x.shape[-2:]
            th, tw = self.size
            if h < th or w < tw:
                raise ValueError("Required crop size {} is larger than input image size {}".format((th, tw), (h, w)))
            if (h, w) == (th, tw):
                return x
            i = int(torch.randint(0, h - th + 1, size=(1,)).item())
            j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
            return x[..., i:i + th, j:j + tw]

    class Pad:
        def __init__(self, padding):
            self.p = padding

        def __call__(self, x):
            return torch.nn.functional.pad(x, (self.p[0], self.p[0], self.p[1], self.p[1]))

    for c in (ColorJitter, Compose, RandomHorizontalFlip, RandomRotation, RandomCrop, Pad):
        setattr(tvt, c.__name__, c)


def loader_train(R):
    """G13: the reference PerspectiveViewLoader's TRAINING item exactly as tasks/pmf/trainer.py:139-142 builds it
    (is_train=True, pcd_aug=False, img_aug=True, use_padding=True), torch seeded per case."""
    from PIL import Image
    _torchvision_transform_standins(sys.modules["torchvision.transforms"])
    out = {}
    for tag, seed, npts, h, w, ht, wt, hp, wp in (("a", 2, 6000, 96, 320, 80, 256, 4, 8),
                                                 ("b", 7, 9000, 64, 208, 64, 208, 2, 4)):
        M, pts, sem, img, lut = loader_ref.synthetic_frame(seed, npts, h, w)
        ds = object.__new__(R.parser.SemanticKitti)
        ds.has_image = True
        ds.proj_matrix = {"00": M}
        ds.class_map_lut = lut
        ds.loadDataByIndex = lambda i: (pts, sem, np.zeros_like(sem))
        ds.loadImage = lambda i: Image.fromarray(img)
        ds.parsePathInfoByIndex = lambda i: ("00", "000000")
        ds.pointcloud_files = [None]
        cfg = {"augmentation": {"img_jitter": [0.4, 0.4, 0.4, 0.1]},
               "sensor": {"proj_h": h, "proj_w": w, "proj_ht": ht, "proj_wt": wt, "h_pad": hp, "w_pad": wp}}
        ld = R.loader.PerspectiveViewLoader(ds, cfg, is_train=True, pcd_aug=False, img_aug=True, use_padding=True)
        for rep in range(2):
            torch.manual_seed(100 * seed + rep)
            feat, mask, label = ld[0]
            out["train.%s.%d" % (tag, rep)] = np.concatenate([feat.numpy(), mask.numpy()[None], label.numpy()[None]], 0)
    np.savez_compressed(os.path.join(OUT, "g13_loader_train.npz"), **out)
    print("g13_loader_train: %d arrays" % len(out))


def epmf_trace(R):
    """G14: two optimisation steps of the EPMF task (tasks/epmf/trainer.py, PMFNet branch, use_mtloss): the reference's
    EPMFNet, MultiTaskLoss(6), FocalSoftmaxLoss, Lovasz_softmax and KLDivLoss driven exactly as trainer.py:376-430
    assembles them ([foc_img, lov_img, per_img, per, foc, lov] -> mt_loss), AdamW over lidar_stream + mt_loss
    (weight_decay = settings.weight_decay, :95-109), SGD-Nesterov over the camera stream; 64x128, bs 2, dropout p=0."""
    import math
    out = {}
    E = _load("refpc.models.epmf_net", "pc_processor/models/epmf_net.py")
    mt = _load("refpc_mtloss", "pc_processor/loss/multi_task_loss.py")
    m = deterministic_init(E.EPMFNet(pcd_channels=5, img_channels=3, nclasses=20, base_channels=32,
                                     imagenet_pretrained=False, image_backbone="resnet34"))
    set_p0(m)
    m.train()
    mtl = mt.MultiTaskLoss(6)
    lr, wd, tau = 0.001, 1e-5, 0.7
    adam = torch.optim.AdamW([{"params": m.lidar_stream.parameters()}, {"params": mtl.parameters()}], lr=lr, weight_decay=wd)
    sgd = torch.optim.SGD([{"params": m.camera_stream_encoder.parameters()},
                           {"params": m.camera_stream_decoder.parameters()}],
                          lr=lr, nesterov=True, momentum=0.9, weight_decay=wd)
    alpha = torch.linspace(0.2, 1.0, 20)
    alpha[0] = 0
    lov = R.lovasz.Lovasz_softmax(ignore=0)
    foc = R.focal.FocalSoftmaxLoss(20, gamma=2, alpha=alpha.numpy(), softmax=False)
    kl = torch.nn.KLDivLoss(reduction="none")
    pcd, rgb, label, _ = synthetic_batch(2, 64, 128, 20, seed=1, fill=0.3)
    label_mask = label.gt(0)
    vals, sig = [], []
    for step in range(2):
        lidar_pred, camera_pred = m(pcd, rgb)
        lidar_pred_log = torch.log(lidar_pred.clamp(min=1e-8))
        pcd_entropy = -(lidar_pred * lidar_pred_log).sum(1) / math.log(20)
        camera_pred_log = torch.log(camera_pred.clamp(min=1e-8))
        img_entropy = -(camera_pred * camera_pred_log).sum(1) / math.log(20)
        pcd_conf, img_conf = 1 - pcd_entropy, 1 - img_entropy
        imp = pcd_conf - img_conf
        pcd_w = imp.gt(0).float() * imp.abs() * pcd_conf.ge(tau).float()
        img_w = imp.lt(0).float() * imp.abs() * img_conf.ge(tau).float()
        loss_per = (kl(lidar_pred_log, camera_pred) * img_w.unsqueeze(1)).mean()
        loss_per_img = (kl(camera_pred_log, lidar_pred) * pcd_w.unsqueeze(1)).mean()
        loss_foc_img = foc(camera_pred, label, mask=label_mask)
        loss_lov_img = lov(camera_pred, label)
        loss_foc = foc(lidar_pred, label, mask=label_mask)
        loss_lov = lov(lidar_pred, label)
        loss_list = [loss_foc_img.unsqueeze(0), loss_lov_img.unsqueeze(0), loss_per_img.unsqueeze(0),
                     loss_per.unsqueeze(0), loss_foc.unsqueeze(0), loss_lov.unsqueeze(0)]
        total = mtl(loss_list)
        adam.zero_grad()
        sgd.zero_grad()
        total.backward()
        if step == 0:
            out["etrace.gsigma0"] = mtl.sigma.grad.double().numpy().copy()
        adam.step()
        sgd.step()
        vals.append([total.item()] + [x.item() for x in loss_list])
        sig.append(mtl.sigma.detach().double().numpy().copy())
        if step == 0:
            sd = m.state_dict()
            for k in ("lidar_stream.downCntx.conv1.conv.weight", "lidar_stream.resBlock3.conv3.weight",
                      "lidar_stream.logits.bias", "camera_stream_encoder.conv1.weight",
                      "camera_stream_decoder.conv.weight", "lidar_stream.extraUpSample.0.weight"):
                out["etrace.param1." + k] = np.array([sd[k].double().sum().item(), sd[k].double().abs().sum().item()])
    out["etrace.losses"] = np.array(vals)            # rows: [total, foc_img, lov_img, per_img, per, foc, lov]
    out["etrace.sigma"] = np.array(sig)
    np.savez_compressed(os.path.join(OUT, "g14_epmf_trace.npz"), **out)
    print("g14_epmf_trace: %d arrays" % len(out))


def nus(R):
    """G15: the reference's NusPerspectiveViewLoader class (tasks/pmf_eval_nuscenes/nus_perspective_loader.py, imported from
    its source file: numpy + torch only) on oracle.cases.SyntheticNus -- the six views of sweep 0 -- and its getMergePred
    (infer.py:18-38, compiled from source as in merge()) on per-view confidences / labels derived from those views.
    Stored per view: depth / xyz / intensity / mask / label planes, the index arrays, and of the three image planes (the
    uint8 image / 255, 154 KB per view) a float64 checksum plus 256 sampled values."""
    import ast
    from oracle.cases import SyntheticNus
    mod = _load("ref_nus_loader", "tasks/pmf_eval_nuscenes/nus_perspective_loader.py")
    ds = SyntheticNus(seed=0, sweeps=2, npts=6000, h=80, w=160, nclasses=17)
    ld = mod.NusPerspectiveViewLoader(dataset=ds, config={})
    out = {}
    rng = np.random.Generator(np.random.PCG64(15))
    samp = rng.integers(0, 3 * 80 * 160, 256)
    idx_l, conf_l, lab_l = [], [], []
    for v in range(6):
        feat, mask, label, xd, yd, dep, pidx, psize = ld[v]
        feat = feat.numpy()
        out["v%d.geom" % v] = feat[:5]
        out["v%d.rgb_sum" % v] = np.array([feat[5:8].astype(np.float64).sum()])
        out["v%d.rgb_samples" % v] = feat[5:8].reshape(-1)[samp]
        out["v%d.mask" % v], out["v%d.label" % v] = mask.numpy(), label.numpy()
        out["v%d.x" % v], out["v%d.y" % v] = xd.numpy(), yd.numpy()
        out["v%d.depth" % v], out["v%d.pidx" % v] = dep.numpy(), pidx.numpy()
        out["v%d.psize" % v] = psize.numpy()
        # a deterministic stand-in for the network: confidence / label of a point from its view and pixel
        conf = ((xd.numpy() * 31 + yd.numpy() * 17 + v * 7) % 97).astype(np.float32) / 97.0
        lab = ((xd.numpy() * 5 + yd.numpy() * 3 + v) % 16 + 1).astype(np.int64)
        idx_l.append(pidx.numpy()), conf_l.append(conf), lab_l.append(lab)
    out["rgb_sample_index"] = samp
    path = os.path.join(REF, "tasks/pmf_eval_nuscenes/infer.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "getMergePred"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        t = lambda xs: [torch.from_numpy(x) for x in xs]
        out["merged"] = ns["getMergePred"](t(idx_l), t(conf_l), t(lab_l), 6000).numpy()
    finally:
        torch.Tensor.cuda = keep
    np.savez_compressed(os.path.join(OUT, "g15_nus.npz"), **out)
    print("g15_nus:", {k: v.shape for k, v in out.items() if k.startswith("v0.") or k == "merged"},
          "unseen points:", int((out["merged"] < 0).sum()), "kept per view:", [out["v%d.x" % v].shape[0] for v in range(6)])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    R = import_reference()
    which = sys.argv[1:] or ["blocks", "whole_net", "losses_metrics", "knn", "loader", "trainer_trace", "epmf", "loader_v2", "range_loader", "kitti_formats", "merge", "loader_train", "epmf_trace", "nus"]
    for name in which:
        globals()[name](R)
