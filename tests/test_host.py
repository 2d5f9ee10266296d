"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the binding's struct
layout matches, the module tree reproduces the reference's state-dict, and plans build (graph logic only)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
import types

from pmf_amd import _lib as L
from pmf_amd.models import PMFNet, SalsaNext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    lib = L.lib()
    hdr = open(os.path.join(ROOT, "include", "pmf_amd.h")).read()
    declared = set(re.findall(r"\b(pmf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libpmf_amd.so does not export %s" % sym
    assert set(L.EXPORTS) <= declared | {"pmf_sizeof", "pmf_version"}
    assert b"gfx950" in lib.pmf_version()


def test_struct_sizes_match_binding():
    lib = L.lib()
    for which, st in enumerate((L.Src, L.ConvDesc, L.WgradDesc, L.View, L.SmallArgs, L.Op, L.PackJob)):
        assert lib.pmf_sizeof(which) == C.sizeof(st)


def test_argument_errors_without_gpu():
    lib = L.lib()
    d = L.ConvDesc()
    assert lib.pmf_conv_fwd(C.byref(d), None) == -1            # nsrc = 0 -> PMF_E_ARG
    assert lib.pmf_knn_vote(None, None, None, None, None, 4, 4, 10, 5, 4, None, C.c_float(1.0), 20, None, None) == -1
    assert lib.pmf_range_project_index(None, 1, 4, 1.0, 1.0, 1.0, 1.0, 4, 4, None, None, None, None, None) == -1
    assert lib.pmf_range_project_gather(None, 1, 4, None, 4, 4, None, None, None, None, None, None, None, None, None,
                                        None) == -1
    assert lib.pmf_points_transform(None, 1, 2, 0, 0, 0.0, 0.0, 0.0, None, None) == -1
    assert lib.pmf_merge_pred(0, None, None, None, None, 4, None, None, None) == -1


def test_range_loader_surface_and_no_cpu_fallback():
    """salsanext_loader.py / projection.py / augmentor.py mirrors: same names, random-draw order, and a loud failure
    (never a CPU path) without a GPU"""
    import random
    import pytest
    from oracle import range_projection_ref as RR
    from oracle.cases import RANGE_CASES
    from pmf_amd.dataset import SalsaNextLoader
    from pmf_amd.dataset.preprocess import augmentor, projection
    cfg = RANGE_CASES[0][3]
    ds = types.SimpleNamespace(loadDataByIndex=lambda i: (np.ones((8, 4), np.float32), np.zeros(8, np.int32), None),
                               labelMapping=lambda l: l, __len__=lambda: 3)
    ld = SalsaNextLoader(ds, cfg, is_train=True, return_uproj=True, device="cpu")
    assert isinstance(ld.projection, projection.RangeProjection) and isinstance(ld.augmentor, augmentor.Augmentor)
    p, a = ld.augmentor.parmas, cfg["augmentation"]
    assert all(getattr(p, k) == a[k] for k in a)
    random.seed(7)
    got = ld.augmentor.draw()
    random.seed(7)
    assert got == RR.draw_augmentation(a, random)
    rp = ld.projection
    assert (np.float32(abs(rp.fov_left)), np.float32(rp.fov_h), np.float32(abs(rp.fov_down)), np.float32(rp.fov_v)) == \
        RR.fov_constants(3., -25., -45, 45)
    with pytest.raises(RuntimeError):
        ld[0]
    with pytest.raises(AssertionError):
        projection.RangeProjection(-1., -25., 512, 64)


def test_state_dict_matches_reference_keys(golden):
    g = golden("g3_wholenet")
    m = PMFNet(imagenet_pretrained=False)
    assert sorted(m.state_dict().keys()) == list(g["r34.keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["r34.nparams"][0])
    m50 = PMFNet(5, 3, 17, 32, False, "resnet50")
    assert sorted(m50.state_dict().keys()) == list(g["r50.keys"])
    assert sum(p.numel() for p in m50.parameters()) == int(g["r50.nparams"][0])
    with pytest.raises(NotImplementedError):
        PMFNet(image_backbone="resnet18", imagenet_pretrained=False)
    # trainer.py:82-89 groups
    n_l = sum(p.numel() for p in m.lidar_stream.parameters())
    n_c = sum(p.numel() for p in m.camera_stream_encoder.parameters()) + \
        sum(p.numel() for p in m.camera_stream_decoder.parameters())
    assert n_l + n_c == 36416040


def test_no_cpu_fallback():
    m = PMFNet(imagenet_pretrained=False)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 5, 32, 64), torch.zeros(1, 3, 32, 64))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 5, 24, 64), torch.zeros(1, 3, 24, 64))
    with pytest.raises(RuntimeError):
        m.lidar_stream.downCntx(torch.zeros(1, 5, 8, 8))


@pytest.mark.parametrize("training", [False, True])
def test_plan_builds_and_is_consistent(training):
    """graph-builder logic on CPU memory (never run): op counts, gradient coverage, descriptor sanity."""
    m = PMFNet(imagenet_pretrained=False).train(training)
    P = m._build(2, 32, 64, training, torch.device("cpu"))
    kinds = [L.OP_NAMES[k] for k in P.fwd_kinds]
    assert kinds.count("OP_CONV") == 110 and kinds.count("OP_SOFTMAX") == 2
    assert kinds.count("OP_BN_FINALIZE" if training else "OP_BN_EVAL") == 94
    for i, k in enumerate(P.fwd_kinds):
        if k == L.OP_CONV:
            d = P.fwd_ops[i].u.conv
            assert (d.w or d.w_s3) and d.out and d.nsrc >= 1 and d.ldw % 64 == 0
            assert sum(d.src[j].C for j in range(d.nsrc)) % 8 == 0
    if training:
        bk = [L.OP_NAMES[k] for k in P.bwd_kinds]
        # (2 x 32 x 64: every map has <= 2048 pixels but the two full-resolution stages; at the bench size the 4x128 stage
        # runs its BatchNorm backward as one launch)
        assert bk.count("OP_WGRAD_PART") == bk.count("OP_WGRAD_RED") == 110
        assert bk.count("OP_BN_BWD_APPLY") + bk.count("OP_BN_BWD_SMALL") == 94
        # scheduling bits: the camera stream on lane 1, the LiDAR stream on lane 0; forward: one event per encoder
        # feature map awaited by the fusion block that reads it; backward: the four fusion blocks' image gradients go to
        # private tensors that the encoder's backward folds in (one ADD_ACT per feature map, waiting for its event)
        fl = [P.fwd_ops[i].pad_ for i in range(P.n_fwd)]
        assert {b & 3 for b in fl} == {0, 1}
        rec = {((b >> 16) & 0xff) - 1 for b in fl if (b >> 16) & 0xff}
        wait = [(((b >> 8) & 0xff) - 1, b & 3) for b in fl if (b >> 8) & 0xff]
        assert len(wait) == 4 and all(l == 0 for _, l in wait) and {e for e, _ in wait} == rec
        bl = [(bk[i], P.bwd_ops[i].pad_) for i in range(P.n_bwd)]
        bwait = [(k, b & 3) for k, b in bl if (b >> 8) & 0xff]
        assert [w for w in bwait if w[1] < 2] == [("OP_ADD_ACT", 1)] * 4
        # weight gradients: batches on lanes 2 / 3, each batch behind one event of its home lane
        side = [w for w in bwait if w[1] >= 2]
        assert side and all(k == "OP_WGRAD_PART" for k, _ in side) and {l for _, l in side} == {2, 3}
        assert all((b & 3) >= 2 for k, b in bl if k in ("OP_WGRAD_PART", "OP_WGRAD_RED"))
        for ops, n in ((P.fwd_ops, P.n_fwd), (P.bwd_ops, P.n_bwd)):      # an event is recorded before it is awaited
            seen = set()
            for i in range(n):
                b = ops[i].pad_
                if (b >> 8) & 0xff:
                    assert ((b >> 8) & 0xff) - 1 in seen
                if (b >> 16) & 0xff:
                    seen.add(((b >> 16) & 0xff) - 1)
        # every parameter has a slot in the flat gradient buffer
        assert {id(p) for p in m.parameters()} == {id(p) for p in P.params}
        offs = sorted((P._pid[id(p)][1], p.numel()) for p in P.params)
        for (o1, n1), (o2, _) in zip(offs, offs[1:]):
            assert o1 + n1 <= o2
        # dropout multiplier table covers the 15 application sites + 3 derived products
        assert len(P.mask_sites) == 15 and len(P.mask_derived) == 3


def _issue_order(ops, n):
    out = (C.c_int32 * n)()
    assert L.lib().pmf_plan_issue_order(C.cast(ops, C.c_void_p), 0, n, C.cast(out, C.c_void_p)) == 0
    return list(out)


@pytest.mark.parametrize("wgrad_lane", ["0", "2", "23"])
def test_plan_issue_order_keeps_lane_order_and_events(wgrad_lane, monkeypatch):
    """pmf_plan_run / pmf_plan_capture hand the ops to the streams in the order of a simulated parallel execution
    (csrc/plan.cpp issue_order): a permutation of the range that keeps every lane's own order, issues the record of an
    event before its wait, and forks a side lane only behind the main-lane ops that precede its first op in the list."""
    monkeypatch.setenv("PMF_WGRAD_LANE", wgrad_lane)
    m = PMFNet(imagenet_pretrained=False).train(True)
    P = m._build(2, 32, 64, True, torch.device("cpu"))
    for ops, n in ((P.fwd_ops, P.n_fwd), (P.bwd_ops, P.n_bwd)):
        order = _issue_order(ops, n)
        assert sorted(order) == list(range(n))
        pos = {k: i for i, k in enumerate(order)}
        lanes = {}
        for k in range(n):
            lanes.setdefault(ops[k].pad_ & 3, []).append(k)
        for lane, ks in lanes.items():                      # per-lane order
            assert [pos[k] for k in ks] == sorted(pos[k] for k in ks)
        rec = {}
        for k in range(n):
            r = ((ops[k].pad_ >> 16) & 0xff) - 1
            if r >= 0:
                rec.setdefault(r, k)
        nwait = 0
        for k in range(n):
            w = ((ops[k].pad_ >> 8) & 0xff) - 1
            if w >= 0 and w in rec and rec[w] < k:
                assert pos[rec[w]] < pos[k]
                nwait += 1
        assert nwait >= 4
        for lane, ks in lanes.items():                      # fork point
            if lane:
                for k0 in lanes[0]:
                    if k0 < ks[0]:
                        assert pos[k0] < pos[ks[0]]
        # the lanes really interleave: the camera lane's ops are not one block behind the LiDAR lane's
        seq = [ops[k].pad_ & 3 for k in order if (ops[k].pad_ & 3) < 2]
        assert sum(1 for a, b in zip(seq, seq[1:]) if a != b) > 40
    if wgrad_lane != "0":
        bl = [P.bwd_ops[i].pad_ & 3 for i in range(P.n_bwd) if P.bwd_kinds[i] == L.OP_WGRAD_PART]
        assert set(bl) <= {2, 3} and (wgrad_lane == "2") == (set(bl) == {2})


def test_salsanext_plan_builds():
    m = SalsaNext(5, 20, 32).eval()
    P = m._build(1, 32, 64, False, torch.device("cpu"))
    assert [L.OP_NAMES[k] for k in P.fwd_kinds].count("OP_CONV") == 51


def test_center_crop_geometry_matches_oracle():
    from oracle import loader_ref
    rng = np.random.default_rng(0)
    x = rng.random((3, 20, 30)).astype(np.float32)
    for (oh, ow, hp, wp) in ((16, 24, 2, 3), (26, 40, 1, 2), (21, 31, 0, 0)):
        ref = loader_ref.center_crop_pad(x, oh, ow, hp, wp)
        assert ref.shape == (3, oh, ow)


def test_semantic_kitti_formats_match_reference(golden, tmp_path):
    """.bin / .label / calib.txt / label config parsing and the prediction writer on a synthetic on-disk tree, against
    what the reference's parser returned for the same tree (tests/golden/g11_kitti_formats.npz)"""
    from oracle.cases import kitti_tree
    from pmf_amd.dataset.semantic_kitti import SemanticKitti, write_prediction
    root = str(tmp_path)
    cfg, data = kitti_tree(root)
    g = golden("g11_kitti_formats")
    ds = SemanticKitti(root, [8, 0], cfg)
    assert len(ds) == 6 and ds.sequences == [0, 8]
    for name in ("class_map_lut", "class_map_lut_inv", "cls_freq", "sem_color_lut", "sem_color_lut_inv"):
        got = getattr(ds, name)
        assert got.dtype == g[name].dtype and np.array_equal(got, g[name]), name
    for name, files in (("order", ds.pointcloud_files), ("label_order", ds.label_files), ("image_order", ds.image_files)):
        assert [os.path.relpath(f, root) for f in files] == list(g[name])
    assert [list(ds.parsePathInfoByIndex(i)) for i in range(6)] == g["path_info"].tolist()
    for seq in ("00", "08"):
        assert np.array_equal(ds.proj_matrix[seq], g["proj." + seq])            # float64 P2 . Tr, bit for bit
    pc, sem, inst = ds.loadDataByIndex(4)
    assert np.array_equal(pc, g["f4.points"]) and np.array_equal(sem, g["f4.sem"]) and np.array_equal(inst, g["f4.inst"])
    assert sem.dtype == np.int32 and np.array_equal(ds.labelMapping(sem), g["f4.mapped"])
    assert np.array_equal(np.asarray(ds.loadImage(4)), g["f4.image"])
    assert np.array_equal(ds.loadLabelByIndex(4)[0], sem)
    # prediction writer: learning ids -> original ids, int32, KITTI submission layout; round trip through readLabel
    path = write_prediction(ds, 4, ds.labelMapping(sem), os.path.join(root, "out"))
    assert path.endswith(os.path.join("sequences", "08", "predictions", "000001.label"))
    back = np.fromfile(path, dtype=np.int32)
    assert np.array_equal(back, ds.class_map_lut_inv[ds.class_map_lut[sem]]) and back.shape == sem.shape
    # without labels: zero labels of the right length; error behaviour of the constructor
    nl = SemanticKitti(root, [0], cfg, has_image=False, has_label=False)
    p0, s0, i0 = nl.loadDataByIndex(0)
    assert s0.shape == (p0.shape[0],) and not s0.any() and not i0.any() and nl.proj_matrix == {}
    with pytest.raises(ValueError):
        SemanticKitti(root, [0], os.path.join(root, "missing.yaml"))
    with pytest.raises(ValueError):
        SemanticKitti(os.path.join(root, "nowhere"), [0], cfg)


def test_pc_processor_shim_resolves_reference_imports():
    """the import statements of tasks/pmf, tasks/epmf and tasks/salsanext resolve to pmf_amd"""
    import importlib
    import pc_processor
    from pc_processor.dataset.preprocess import augmentor, projection
    from pc_processor.dataset.semantic_kitti import SemanticKitti
    import pmf_amd
    assert pc_processor.dataset.SalsaNextLoader is pmf_amd.dataset.SalsaNextLoader
    assert importlib.import_module("pc_processor.dataset.salsanext_loader").SalsaNextLoader is pmf_amd.dataset.SalsaNextLoader
    assert projection.RangeProjection is pmf_amd.dataset.preprocess.projection.RangeProjection
    assert augmentor.Augmentor is pmf_amd.dataset.preprocess.augmentor.Augmentor
    assert SemanticKitti is pmf_amd.dataset.semantic_kitti.SemanticKitti
    for name in ("PMFNet", "EPMFNet", "SalsaNext"):
        assert getattr(pc_processor.models, name) is getattr(pmf_amd.models, name)
    for name in ("Lovasz_softmax", "FocalSoftmaxLoss", "MultiTaskLoss"):
        assert hasattr(pc_processor.loss, name)
    assert pc_processor.postproc.KNN is pmf_amd.postproc.KNN and hasattr(pc_processor.utils, "WarmupCosineLR")


def test_tensor_augmentation_restatement_and_draw_order():
    """oracle/tensor_aug_ref.py: documented torchvision behaviour (counter-clockwise for positive angles, identity at 0,
    zero fill) and the same torch-RNG draw order in the product's FlipRotateCrop"""
    from oracle import tensor_aug_ref as T
    from pmf_amd.dataset import FlipRotateCrop
    img = torch.arange(2 * 6 * 6, dtype=torch.float32).reshape(2, 6, 6)
    assert torch.equal(T.rotate_nearest(img, 0.0), img)
    assert torch.equal(T.rotate_nearest(img, 90.0), torch.rot90(img, 1, (1, 2)))
    assert torch.equal(T.flip_rotate_crop(img, True, 0.0, 1, 2, 3, 4, 1, 2)[:, 1:4, 2:6], img.flip(-1)[:, 1:4, 2:6])
    big = T.rotate_nearest(torch.ones(1, 40, 64), 15.0)
    assert big[0, 0, 0] == 0 and big[0, 20, 32] == 1                     # corners leave the frame, centre stays
    op = FlipRotateCrop(32, 48, 2, 3)
    torch.manual_seed(11)
    a = [op.draw(40, 64) for _ in range(5)]
    torch.manual_seed(11)
    b = [T.draw_params(40, 64, 32, 48) for _ in range(5)]
    assert a == b and any(f for f, _, _, _ in a) and not all(f for f, _, _, _ in a)
    assert all(-15 <= ang <= 15 and 0 <= t <= 8 and 0 <= l <= 16 for _, ang, t, l in a)
    assert op.draw(32, 48)[2:] == (0, 0)
    with pytest.raises(ValueError):
        op.draw(16, 64)
    with pytest.raises(RuntimeError):
        op(torch.zeros(10, 40, 64))                                      # CPU tensor: no fallback
    assert L.lib().pmf_flip_rotate_crop(None, 10, 4, 4, 0, None, 0, 0, 4, 4, 0, 0, None, 4, 4, None) == -1


def test_prefetcher_stops_with_its_consumer():
    """leaving the loop early must end the producer thread (it used to block in q.put forever, holding
    `depth` batches and drawing from the global RNG next to the following epoch's producer)."""
    import threading
    import time
    from tasks.pmf.trainer import Prefetcher
    produced = []

    class Src(object):
        def __len__(self):
            return 1000

        def __iter__(self):
            for i in range(1000):
                produced.append(i)
                yield i
    before = threading.active_count()
    pf = Prefetcher(Src(), depth=2)
    got = []
    for x in pf:
        got.append(x)
        if x == 3:
            break
    t0 = time.time()
    while threading.active_count() > before and time.time() - t0 < 5.0:
        time.sleep(0.01)
    assert got == [0, 1, 2, 3]
    assert threading.active_count() == before, "producer thread still alive after the consumer left"
    assert len(produced) < 20                 # it did not run on through the source
    assert list(Prefetcher(Src(), depth=2)) == list(range(1000))      # and a full pass still delivers everything in order

    class Boom(object):
        def __iter__(self):
            yield 1
            raise ValueError("boom")
    with pytest.raises(ValueError):
        list(Prefetcher(Boom()))


def test_prefetcher_workers_keep_sampler_order():
    """several prefetch threads build alternate batches of a DataLoader's batch sampler; delivery stays in sampler order and
    every item arrives exactly once (also when the batch count is not a multiple of the worker count)"""
    import torch
    from torch.utils.data import DataLoader, Dataset
    from tasks.pmf.trainer import Prefetcher

    class DS(Dataset):
        def __len__(self):
            return 23

        def __getitem__(self, i):
            return torch.tensor([i, 2 * i])
    dl = DataLoader(DS(), batch_size=3, shuffle=False, drop_last=False)
    want = [b.tolist() for b in dl]
    for w in (2, 3):
        got = [b.tolist() for b in Prefetcher(dl, depth=2, workers=w)]
        assert got == want, (w, got)
    it = iter(Prefetcher(dl, workers=3))
    assert next(it).tolist() == want[0]
    del it                                     # early exit with several workers: threads end (no hang at interpreter exit)


def test_documented_knob_defaults_are_what_a_fresh_plan_uses(monkeypatch):
    """docs/knobs.md lists the PMF_* knobs and says "defaults are what the bench runs": a plan built in a clean
    environment must carry exactly these values, and every knob named here must be documented there."""
    from pmf_amd.plan import Plan
    for k in list(os.environ):
        if k.startswith("PMF_"):
            monkeypatch.delenv(k)
    got = Plan(torch.device("cpu"), True).knobs()
    want = {"PMF_CONV_F32": False, "PMF_S3_MIN_TAPS": 2, "PMF_S3_DIRECT_MIN_PIX": 1, "PMF_BN_BWD_FUSED": True, "lanes": 4,
            "PMF_WGRAD_LANE": 23, "PMF_WGRAD_BATCH": 4,
            "PMF_RED_BATCH": 32, "PMF_BN_SMALL": True, "PMF_DGRAD_MERGE": True, "PMF_DGRAD_MERGE_MINPIX": 1024,
            "PMF_AUTOTUNE": True, "PMF_TUNE_DIRECT": True, "PMF_GRAPH": True, "PMF_DP_MODE": "events", "PMF_DP_SEGMENTS": 4,
            "PMF_PACK_EARLY": 8}
    assert got == want
    design = open(os.path.join(ROOT, "docs", "knobs.md")).read()
    for k in want:
        if k.startswith("PMF_"):
            assert k in design, "%s is not documented in docs/knobs.md" % k
    # ... and every PMF_* variable the product sources READ is listed there (no undocumented code paths)
    import glob
    import re
    read = set()
    for f in glob.glob(os.path.join(ROOT, "pmf_amd", "**", "*"), recursive=True) + [os.path.join(ROOT, "bench.py")]:
        if f.endswith((".py", ".hip", ".cpp", ".h")):
            src = open(f).read()
            read |= set(re.findall(r'getenv\("(PMF_[A-Z0-9_]+)"', src))
            read |= set(re.findall(r'environ(?:\.get|\.pop|\.setdefault)?[\[(]"(PMF_[A-Z0-9_]+)"', src))
    undocumented = sorted(k for k in read if k not in design)
    assert not undocumented, undocumented
