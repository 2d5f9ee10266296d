"""CPU: oracle/act_masks.py -- the decision recorder / injector the masked backward parity check is built on
(bench.py --parity-masked, tests/test_gpu_fullsize.py::test_masked_backward_parity).  Reference activations:
pc_processor/models/salsanext.py:27-33, pmf_net.py:20-29,94."""
import copy

import pytest
import torch

from oracle import epmf_torch as E
from oracle import pmf_torch as O
from oracle.act_masks import ActSites
from pmf_amd.utils.detinit import deterministic_init, synthetic_batch


def _nets():
    return {"pmf_r34": lambda: O.PMFNet(5, 3, 20, 32, False, "resnet34"),
            "pmf_r50": lambda: O.PMFNet(5, 3, 17, 32, False, "resnet50"),
            "epmf": lambda: E.EPMFNet(5, 3, 20, 32, False, "resnet34")}


@pytest.mark.parametrize("kind", ["pmf_r34", "pmf_r50", "epmf"])
def test_own_decisions_reproduce_the_pass_bit_for_bit(kind):
    net = deterministic_init(_nets()[kind]()).train()
    pcd, rgb, _, _ = synthetic_batch(1, 32, 64, 20, seed=1, fill=0.3)
    a_net, b_net, c_net = copy.deepcopy(net), copy.deepcopy(net), copy.deepcopy(net)
    torch.manual_seed(0)
    a, b = a_net(pcd, rgb)
    (a.square().sum() + b.square().sum()).backward()
    with ActSites(b_net) as rec:
        torch.manual_seed(0)
        b_net(pcd, rgb)
    kinds = {k[0] for k in rec.decisions}
    assert kinds == {"lrelu", "relu", "relu_out", "maxpool"}
    # one site per activation module of the reference tree: every conv of the LiDAR stream but logits / ASPP, every
    # BatchNorm-ReLU of the camera encoder, every residual sum, the stem pool
    assert ("maxpool", "camera_stream_encoder.conv1") in rec.decisions
    assert ("relu_out", "camera_stream_encoder.layer4.2") in rec.decisions
    assert ("relu", "lidar_stream.fusionblock_1.attention.0") in rec.decisions
    with ActSites(c_net, inject=rec.decisions) as inj:
        torch.manual_seed(0)
        a2, b2 = c_net(pcd, rgb)
    assert not inj.unused
    (a2.square().sum() + b2.square().sum()).backward()
    assert torch.equal(a, a2) and torch.equal(b, b2)
    for p, q in zip(a_net.parameters(), c_net.parameters()):
        assert torch.equal(p.grad, q.grad)
    # the functionals are restored
    import torch.nn.functional as F
    assert F.relu.__module__ == "torch.nn.functional" and F.leaky_relu.__module__ == "torch.nn.functional"


def test_injected_decisions_make_two_precisions_differentiate_one_function():
    """flip a handful of decisions by hand: the float64 pass with the injected decisions follows THEM, not its own signs --
    its gradient moves exactly as the fp32 pass's with the same decisions does"""
    net = deterministic_init(_nets()["pmf_r34"]()).train()
    pcd, rgb, _, _ = synthetic_batch(1, 32, 64, 20, seed=2, fill=0.3)
    g = torch.Generator().manual_seed(3)
    masks = {nm: (torch.rand(1, c, generator=g) > 0.2).float() / 0.8 for nm, _, c in O.dropout_sites(net)}
    base = copy.deepcopy(net)           # (a forward pass leaves non-leaf tensors on the module: copy first)
    with ActSites(net) as rec:
        O.set_dropout_masks(net, masks)
        net(pcd, rgb)
    net = base
    dec = dict(rec.decisions)
    k = ("lrelu", "lidar_stream.upBlock4.conv2")
    flipped = dec[k].clone()
    flipped[0, :, 5:9, 5:9] = ~flipped[0, :, 5:9, 5:9]
    dec[k] = flipped
    grads = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = copy.deepcopy(net).to(dt)
        O.set_dropout_masks(m, {kk: v.to(dt) for kk, v in masks.items()})
        with ActSites(m, inject=dec):
            a, b = m(pcd.to(dt), rgb.to(dt))
        (a.square().sum() + b.square().sum()).backward()
        grads[tag] = {n: p.grad.double() for n, p in m.named_parameters()}
    m = copy.deepcopy(net).double()
    O.set_dropout_masks(m, {kk: v.double() for kk, v in masks.items()})
    a, b = m(pcd.double(), rgb.double())
    (a.square().sum() + b.square().sum()).backward()
    own = {n: p.grad for n, p in m.named_parameters()}
    n = "lidar_stream.upBlock4.conv1.weight"
    d_inj = float((grads["f32"][n] - grads["f64"][n]).norm() / grads["f64"][n].norm())
    d_own = float((grads["f32"][n] - own[n]).norm() / own[n].norm())
    assert d_inj < 1e-4 < d_own, (d_inj, d_own)
