"""Host logic (CPU): the product loss modules (device-agnostic torch ops) against the reference-run fixture."""
import numpy as np
import torch

from pmf_amd.loss import FocalSoftmaxLoss, Lovasz_softmax, pmf_total_loss
from pmf_amd.utils.detinit import det_tensor, synthetic_batch


def test_losses_match_reference_fixture(golden):
    g = golden("g6_losses")
    n, c, h, w = 2, 20, 16, 32
    a = det_tensor("g6.logits", (n, c, h, w), -3, 3).requires_grad_(True)
    b = det_tensor("g6.logits2", (n, c, h, w), -3, 3).requires_grad_(True)
    _, _, label, _ = synthetic_batch(n, h, w, c, seed=3, fill=0.4)
    alpha = np.linspace(0.2, 1.0, c).astype(np.float32)
    alpha[0] = 0
    foc = FocalSoftmaxLoss(c, gamma=2, alpha=alpha, softmax=False)
    lov = Lovasz_softmax(ignore=0)
    total, t = pmf_total_loss(torch.softmax(a, 1), torch.softmax(b, 1), label, foc, lov)
    total.backward()
    vals = np.array([total.item()] + [t[k].item() for k in ("foc", "lov", "foc_cam", "lov_cam", "per")])
    assert np.abs(vals - g["loss.values"]).max() < 2e-6
    assert np.abs(a.grad.numpy() - g["loss.grad_a"]).max() < 2e-7
    assert np.abs(b.grad.numpy() - g["loss.grad_b"]).max() < 2e-7


def test_lovasz_edge_cases():
    lov = Lovasz_softmax(ignore=0)
    p = torch.softmax(torch.randn(1, 5, 4, 4), 1).requires_grad_(True)
    out = lov(p, torch.zeros(1, 4, 4, dtype=torch.long))        # only void pixels -> 0, zero gradients
    out.backward()
    assert out.item() == 0 and p.grad.abs().max() == 0
    lab = torch.full((1, 4, 4), 3, dtype=torch.long)              # single present class
    assert torch.isfinite(lov(p, lab))


def test_engine_two_steps_match_reference_trace(golden):
    """G7: TrainEngine (AdamW lidar / SGD-Nesterov camera grouping, loss weights) against two optimisation steps driven
    with the reference's modules (tests/golden/g7_trace.npz); the CPU oracle network stands in for the HIP model."""
    from oracle import pmf_torch as O
    from pmf_amd.engine import TrainEngine
    from pmf_amd.utils.detinit import deterministic_init
    g = golden("g7_trace")
    m = deterministic_init(O.PMFNet(5, 3, 20, 32, False, "resnet34"))
    for x in m.modules():
        if isinstance(x, O.DropSite):
            x.p = 0.0
    alpha = np.linspace(0.2, 1.0, 20).astype(np.float32)
    alpha[0] = 0
    eng = TrainEngine(m, 20, lr=1e-3, momentum=0.9, weight_decay=1e-5, alpha=alpha, warmup_steps=1, max_steps=10 ** 9)
    pcd, rgb, label, mask = synthetic_batch(1, 64, 512, 20, seed=1)
    feat = torch.cat((pcd, rgb), 1)
    vals = []
    for _ in range(2):
        total, t = eng.train_step(feat.clone(), torch.ones_like(mask), label)
        vals.append([total.item()] + [t[k].item() for k in ("foc", "lov", "foc_cam", "lov_cam", "per")])
    assert np.abs(np.array(vals) - g["trace.losses"]).max() < 5e-4 * np.abs(g["trace.losses"]).max()
    sd = m.state_dict()
    for k in [k for k in g.files if k.startswith("trace.param.")]:
        name = k[len("trace.param."):]
        got = np.array([sd[name].double().sum().item(), sd[name].double().abs().sum().item()])
        assert np.abs(got - g[k]).max() <= 1e-3 * max(np.abs(g[k]).max(), 1e-3), name


def test_multitask_loss_matches_reference_fixture(golden):
    """pmf_amd.loss.MultiTaskLoss (pc_processor/loss/multi_task_loss.py:5-19) vs the reference-run values in g8_epmf."""
    import numpy as np
    import torch
    from pmf_amd.loss import MultiTaskLoss
    g = golden("g8_epmf")
    mtl = MultiTaskLoss(6)
    ls = [torch.tensor(v, requires_grad=True) for v in (0.7, 1.3, 0.2, 2.1, 0.9, 0.05)]
    tot = mtl(ls)
    tot.backward()
    assert abs(tot.item() - float(g["mtl.total"][0])) < 1e-6
    assert np.abs(mtl.sigma.grad.numpy() - g["mtl.gsigma"]).max() < 1e-5
    assert np.abs(np.array([l.grad.item() for l in ls]) - g["mtl.gloss"]).max() < 1e-6


def test_epmf_surface_and_no_cpu_fallback():
    """EPMFNet keeps the reference's constructor / attribute surface and state-dict keys, and refuses CPU tensors."""
    import numpy as np
    import pytest
    import torch
    import pc_processor
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g8_epmf.npz"))
    m = pc_processor.models.EPMFNet(pcd_channels=5, img_channels=3, nclasses=20, base_channels=32,
                                    imagenet_pretrained=False, image_backbone="resnet34")
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"][0])
    for attr in ("lidar_stream", "camera_stream_encoder", "camera_stream_decoder"):
        assert len(list(getattr(m, attr).parameters())) > 0
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 5, 64, 64), torch.zeros(1, 3, 64, 64))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 5, 48, 64), torch.zeros(1, 3, 48, 64))
    with pytest.raises(NotImplementedError):
        pc_processor.models.EPMFNet(image_backbone="vgg16")


def test_epmf_engine_two_steps_match_reference_trace(golden):
    """G14: EPMFEngine (six terms through MultiTaskLoss in the reference's order, AdamW over lidar stream + sigmas with
    weight_decay, SGD-Nesterov camera) against two optimisation steps driven with the reference's modules; the CPU
    oracle network stands in for the HIP model (the GPU test replays the same fixture on the HIP model)."""
    from oracle import epmf_torch as E
    from oracle import pmf_torch as O
    from pmf_amd.engine import EPMFEngine
    from pmf_amd.loss import EPMF_TERMS
    from pmf_amd.utils.detinit import deterministic_init
    g = golden("g14_epmf_trace")
    m = deterministic_init(E.EPMFNet(5, 3, 20, 32, False, "resnet34"))
    for x in m.modules():
        if isinstance(x, O.DropSite):
            x.p = 0.0
    alpha = np.linspace(0.2, 1.0, 20).astype(np.float32)
    alpha[0] = 0
    eng = EPMFEngine(m, 20, lr=1e-3, momentum=0.9, weight_decay=1e-5, alpha=alpha, warmup_steps=1, max_steps=10 ** 9)
    assert len(eng.optimizer.param_groups) == 2 and eng.optimizer.param_groups[1]["params"][0] is eng.mt_loss.sigma
    assert eng.optimizer.param_groups[0]["weight_decay"] == 1e-5
    pcd, rgb, label, mask = synthetic_batch(2, 64, 128, 20, seed=1, fill=0.3)
    feat = torch.cat((pcd, rgb), 1)
    for step in range(2):
        total, t = eng.train_step(feat.clone(), torch.ones_like(mask), label)
        got = np.array([total.item()] + [t[k].item() for k in EPMF_TERMS])
        want = g["etrace.losses"][step]
        assert np.abs(got - want).max() < 1e-3 * np.abs(want).max(), (step, got, want)
        assert np.abs(eng.mt_loss.sigma.detach().double().numpy() - g["etrace.sigma"][step]).max() < 2e-6
        if step == 0:
            sd = m.state_dict()
            for k in [k for k in g.files if k.startswith("etrace.param1.")]:
                name = k[len("etrace.param1."):]
                gotc = np.array([sd[name].double().sum().item(), sd[name].double().abs().sum().item()])
                assert np.abs(gotc - g[k]).max() <= 1e-4 * max(abs(g[k][1]), 1e-3), name
