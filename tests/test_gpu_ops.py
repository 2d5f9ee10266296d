"""-m gpu: every plan primitive (forward AND backward kernels) in isolation against float64 torch autograd.

Whole-network gradient comparisons are limited by LeakyReLU/ReLU derivative discontinuities (one pre-activation
within rounding distance of 0 changes a weight gradient by ~1e-3 relative), so the tight kernel-level bars live
here, on small graphs built with the same Plan API the models use."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pmf_amd import _lib as L  # noqa: E402
from pmf_amd.plan import Plan, V  # noqa: E402
from pmf_amd.utils.detinit import det_tensor  # noqa: E402

DEV = "cuda"
TOL = 3e-5


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


class Harness:
    """inputs -> graph(P, views) -> outputs; runs fwd+bwd on the GPU and the same function in float64 autograd."""

    def __init__(self, shapes, needs_grad=None, masks=0):
        self.P = Plan(torch.device(DEV), True)
        self.P.masks = torch.ones(max(masks, 4), device=DEV)
        self.shapes = shapes
        self.inputs = []
        for i, (n, c, h, w) in enumerate(shapes):
            t = self.P.input_nchw("in%d" % i, n, c, h, w, "in%d" % i)
            t.needs_grad = True if needs_grad is None else needs_grad[i]
            self.inputs.append(t)

    def run(self, outs, xs, gouts, params=(), cfg=0):
        """outs: list of V (plan outputs); xs: cpu float32 inputs; gouts: cpu float32 upstream grads (NCHW)."""
        P = self.P
        gts = [P.external_grad(o) for o in outs]
        # gradient inputs are copied in before the backward pass: emit nothing, just remember buffers
        P.finalise()
        if cfg:
            P.force_conv_cfg(cfg)
        for i, x in enumerate(xs):
            xc = x.to(DEV).contiguous()
            a = P.fwd_ops[P.in_slots["in%d" % i] + P.fwd_shift].u.sm
            a.p[0] = xc.data_ptr()
            a.l[0], a.l[1] = xc.stride(0), xc.stride(1)
            setattr(self, "_keep%d" % i, xc)
        P.run(P.fwd_ops, P.n_fwd, "forward")
        torch.cuda.synchronize()
        got = [P.read_view(o).cpu() for o in outs]
        for gt, g in zip(gts, gouts):
            buf = gt.buf.tensor((gt.N, gt.H, gt.W, gt.ldc))
            buf.zero_()
            buf[..., :gt.C] = g.permute(0, 2, 3, 1).to(DEV)
        # the prologue of the backward plan zeroes zero_bwd (param grads, BN scratch); external grads live in `act`
        P.run(P.bwd_ops, P.n_bwd, "backward")
        torch.cuda.synchronize()
        gin = [P.read(t.g).cpu() if (t.needs_grad and t.g is not None) else None for t in self.inputs]
        flat = P.pgrad_buf.tensor((P.pgrad_floats,))
        gp = {}
        for p in params:
            off = P._pid[id(p)][1]
            gp[id(p)] = flat[off:off + p.numel()].view(p.shape).cpu().clone()
        return got, gin, gp


def _check(name, got, ref, tol=TOL):
    e = _err(got, ref)
    assert e < tol, "%s: rel L2 err %.3e" % (name, e)


CONVS = [
    # name, N,H,W, [Cin], Cout, k, dil, pad, stride, bias, order/act, bn
    ("c3x3", 2, 12, 40, [32], 32, 3, 1, 1, 1, True, "act_bn", True),
    ("c3x3d2", 1, 16, 36, [64], 64, 3, 2, 2, 1, True, "act_bn", True),
    ("c2x2d2", 2, 8, 32, [32], 32, 2, 2, 1, 1, True, "act_bn", True),
    ("c1x1cat3", 1, 8, 32, [64, 64, 64], 64, 1, 1, 0, 1, True, "act_bn", True),
    ("c3x3cat2", 1, 8, 64, [16, 64], 32, 3, 1, 1, 1, True, "act_bn", True),
    ("c1x1_plain_lrelu", 2, 8, 32, [16], 32, 1, 1, 0, 1, True, "lrelu", False),
    ("c3x3_plain", 1, 8, 32, [32], 24, 3, 1, 1, 1, True, "none", False),
    ("c7x7", 1, 12, 40, [8], 64, 7, 1, 3, 1, False, "bn_relu", True),
    ("c3x3s2", 2, 16, 32, [64], 128, 3, 1, 1, 2, False, "bn_relu", True),
    ("c1x1s2", 2, 16, 32, [64], 128, 1, 1, 0, 2, False, "bn_norelu", True),
    ("c3x3d6", 1, 8, 40, [64], 64, 3, 6, 6, 1, True, "none", False),
    ("c3x3d18", 1, 8, 40, [32], 32, 3, 18, 18, 1, True, "none", False),
    ("c3x3_big", 2, 4, 16, [256], 256, 3, 1, 1, 1, True, "act_bn", True),
    ("c1x1_512_cat", 1, 4, 16, [256, 512], 256, 3, 1, 1, 1, True, "act_bn", True),
    # >= 16384 pixels and >= 96 input channels: the LDS-free 1x1 weight-gradient kernel (operands straight from global)
    ("c1x1cat3_big", 2, 64, 128, [64, 64, 64], 64, 1, 1, 0, 1, True, "act_bn", True),
    ("c1x1_odd_big", 1, 129, 129, [96], 32, 1, 1, 0, 1, True, "lrelu", False),
    ("c1x1_c20_big", 1, 129, 129, [128], 20, 1, 1, 0, 1, True, "none", False),
    ("c1x1_wide_big", 2, 64, 128, [128, 256], 128, 1, 1, 0, 1, True, "act_bn", True),
    # split-bf16 direct 1x1 weight gradient (64 x 64 blocks, 16-pixel groups): ragged last group, 80 output channels
    ("c1x1_odd_wide", 1, 129, 129, [64, 128], 80, 1, 1, 0, 1, False, "bn_relu", True),
    # shapes of the pipelined weight-gradient kernel (rows % 4 == 0, cols % 32 == 0, channels % 32 == 0), 9 / 4 / 1 taps
    ("c3x3_wpipe", 2, 8, 64, [32, 64], 64, 3, 1, 1, 1, True, "act_bn", True),
    ("c3x3d2_wpipe", 1, 12, 32, [64], 32, 3, 2, 2, 1, False, "bn_relu", True),
    ("c2x2d2_wpipe", 2, 8, 64, [64], 64, 2, 2, 1, 1, True, "act_bn", True),
    # 128 / 256 output channels: the N-split weight-gradient kernel with FOUR output-channel tiles per workgroup (every wave sees
    # all eight slabs of a tile); dilated = the nine-slot input tile; a 16-channel operand = a half-empty chunk
    ("c3x3_wn4", 2, 8, 64, [64], 128, 3, 1, 1, 1, True, "act_bn", True),
    ("c3x3d2_wn4", 1, 12, 32, [32, 16], 256, 3, 2, 2, 1, False, "bn_relu", True),
    ("c2x2d2_wn4", 1, 16, 32, [64], 128, 2, 2, 1, 1, True, "act_bn", True),
    # ResNet-50 bottleneck 1x1 layers at their real channel counts: 1024 / 2048 input channels (the weight fragments of one
    # output-channel tile do not fit LDS: these run on the generic loop), 256 -> 1024, the stride-2 projection
    ("c1x1_k1024", 2, 8, 32, [1024], 256, 1, 1, 0, 1, False, "bn_relu", True),
    ("c1x1_k2048", 1, 4, 64, [2048], 512, 1, 1, 0, 1, False, "bn_relu", True),
    ("c1x1_to1024", 2, 8, 32, [256], 1024, 1, 1, 0, 1, False, "bn_norelu", True),
    ("c1x1_k1040_cat", 1, 8, 32, [1024, 16], 64, 1, 1, 0, 1, True, "act_bn", True),
    ("c1x1s2_k1024", 2, 8, 32, [1024], 2048, 1, 1, 0, 2, False, "bn_norelu", True),
    # stride-2 1x1 on the LDS-free split weight-gradient kernel (round 6: >= 4096 output pixels, or >= 1024 with >= 512 input
    # channels; the ResNet downsample projections): even and odd input maps, operand behind a BatchNorm + ReLU view
    ("c1x1s2_direct", 2, 64, 128, [64], 128, 1, 1, 0, 2, False, "bn_norelu", True),
    ("c1x1s2_direct_odd", 1, 63, 131, [128], 64, 1, 1, 0, 2, True, "act_bn", True),
    ("c1x1s2_direct_k512", 2, 32, 64, [512], 128, 1, 1, 0, 2, False, "bn_relu", True),
    # 768 / 784 input channels: the direct variant streams the weight fragments in chunks through two LDS buffers (two
    # workgroups per CU); ragged last chunk
    ("c1x1_k768_cat", 2, 16, 64, [256, 256, 256], 256, 1, 1, 0, 1, True, "act_bn", True),
    ("c1x1_k784", 1, 16, 64, [784], 64, 1, 1, 0, 1, False, "bn_relu", True),
    # round 5, the weight-gradient tails.  One-tap layers with one or two 32 x 32 output tiles per chunk (logits 32 -> 20,
    # downCntx.s 5 -> 32, resBlock1.s 32 -> 64): the waves of the unit-dealing kernel split the ROWS of a tile and fold at the
    # end (ragged tile columns: 40 / 72 wide).  3x3 layers with 20 / 16 output channels (dec.logits, dec.up2-4): the N-split
    # split-bf16 kernel with a ragged last output-channel tile (dz pitch 24 / 16: lanes beyond Cout read the next pixel)
    ("c1x1_k32_c20", 2, 16, 72, [32], 20, 1, 1, 0, 1, True, "none", False),
    ("c1x1_k8_c32", 2, 8, 64, [8], 32, 1, 1, 0, 1, True, "lrelu", False),
    ("c1x1_k32_c64", 1, 12, 40, [32], 64, 1, 1, 0, 1, True, "act_bn", True),
    ("c3x3_c16_to20", 2, 8, 64, [16], 20, 3, 1, 1, 1, True, "none", False),
    ("c3x3_cat_to16", 1, 8, 32, [128, 16], 16, 3, 1, 1, 1, False, "bn_relu", True),
    # >= 32768 pixels, <= 96 input and <= 64 output channels, one tap: the streaming weight-gradient kernel (operands as dword
    # loads in the fp32 MFMA layout, a workgroup inside one sample); odd pixel count, operands behind BatchNorm / Dropout views
    ("c1x1_stream_k32_c20", 2, 64, 256, [32], 20, 1, 1, 0, 1, True, "none", False),
    ("c1x1_stream_cat", 1, 129, 257, [64, 16], 16, 1, 1, 0, 1, True, "act_bn", True),
    ("c1x1_stream_k8_c64", 2, 64, 256, [8], 64, 1, 1, 0, 1, True, "act_bn", True),
    ("c1x1_stream_k96_c32", 2, 128, 129, [32, 32, 32], 32, 1, 1, 0, 1, False, "bn_relu", True),
    # S_B's low-resolution maps (60 x 80, 30 x 40): partial 4 x 32-pixel tiles on the N-split kernel (dz loads beyond the map
    # take the hardware zero); c3x3 (12 x 40) and c3x3d2 (16 x 36) above run the same instantiations with one / two tiles
    ("c3x3_rag_n4", 2, 30, 40, [64], 128, 3, 1, 1, 1, True, "act_bn", True),
    ("c2x2d2_rag", 1, 15, 80, [32, 32], 64, 2, 2, 1, 1, False, "bn_relu", True),
    # EPMF's first layers: 3x3 over 5 (padded 8) input channels -- a quarter-full chunk on the N-split weight-gradient kernel
    ("c3x3_k8_c32", 2, 16, 64, [8], 32, 3, 1, 1, 1, True, "lrelu", False),
    ("c3x3_k8_cat", 1, 8, 64, [8, 32], 64, 3, 1, 1, 1, True, "act_bn", True),
]


TILE_CFGS = [32 | (1 << 8) | (1 << 16) | L.CFG_WS | (code << 26) for code in (0, 2, 3)] + \
            [bn | (2 << 8) | (1 << 16) for bn in (32, 64)] + \
            [bn | (1 << 8) | (ks << 16) for bn in (32, 64) for ks in (1, 2, 4, 8, 16)] + \
            [bn | (mt << 8) | (1 << 16) | L.CFG_DIRECT_TAPS for bn in (32, 64) for mt in (1, 2)]     # direct multi-tap variant


@pytest.mark.parametrize("cfg", TILE_CFGS, ids=["bn%d_mt%d_ks%d%s" % (c & 255, (c >> 8) & 255, (c >> 16) & 255,
                                                                        "_direct" if (c >> 24) & 1 else ("_ws%d" % ((c >> 26) & 3) if (c >> 25) & 1 else ""))
                                              for c in TILE_CFGS])
def test_conv_tile_configs(cfg):
    """every tile configuration the plan autotuner may write into a conv op (output-channel tile 32/64, 128/256-pixel
    tile, 1-16 K splits), forward + input gradient + BatchNorm statistics, against float64 -- on a 3x3 dilated conv
    with BN, a 3-operand 1x1 (64-channel stages), a stride-2 3x3 (parity-class input gradients) and a deep small map; plus
    the wave-scheduled N-split kernel with its output-tile cap (cfg bits 25-27: default rule / two / four tiles)"""
    for case in CONVS:
        if case[0] in ("c3x3d2", "c1x1cat3", "c3x3s2", "c3x3_big", "c7x7") or \
                (cfg >> 24 and case[0] in ("c3x3", "c2x2d2", "c3x3cat2", "c1x1_512_cat", "c3x3_wpipe", "c2x2d2_wpipe", "c3x3d6")):
            _conv_case(case, cfg)


@pytest.mark.parametrize("case", CONVS, ids=[c[0] for c in CONVS])
def test_conv_unit_fwd_bwd(case):
    _conv_case(case, 0)


@pytest.mark.parametrize("name", ["c3x3", "c3x3d2", "c1x1cat3", "c3x3_big", "c3x3s2", "c2x2d2_wpipe"])
def test_conv_unit_bn_backward_three_launch_form(name, monkeypatch):
    """maps of <= 2048 pixels run their BatchNorm backward as ONE launch (bn_bwd_small_k) by default -- which is what the
    cases above exercise; PMF_BN_SMALL=0 keeps them on the reduce / fold / apply kernels and on the input-gradient epilogue
    that carries the column sums, the form every larger map uses"""
    monkeypatch.setenv("PMF_BN_SMALL", "0")
    _conv_case(next(c for c in CONVS if c[0] == name), 0)


@pytest.mark.parametrize("name", ["c3x3_big", "c3x3d2", "c1x1cat3", "c3x3s2"])
def test_conv_unit_splitk_two_launch_form(name, monkeypatch):
    """split-K launches combine their partial slabs inside the kernel by default (pmf_conv_desc_t.splitk_tickets, round 6) -- what
    test_conv_tile_configs' ks > 1 configurations exercise; PMF_SPLITK_FUSED=0 keeps them on the second launch (conv_finish_k):
    forward, input gradients and BatchNorm statistics of both forms against float64"""
    monkeypatch.setenv("PMF_SPLITK_FUSED", "0")
    for ks in (2, 8):
        _conv_case(next(c for c in CONVS if c[0] == name), 32 | (1 << 8) | (ks << 16))


@pytest.mark.parametrize("env", [{"PMF_WG_SWP": "0"}, {"PMF_WG_W8": "1"}, {"PMF_WG_S3N": "0"}, {"PMF_WG_S3N": "2"}],
                         ids=["staged", "eight_waves", "swp_pixel_split", "nsplit_two_tiles"])
def test_conv_unit_wgrad_variants(env, monkeypatch):
    """the split-bf16 weight-gradient kernel's other forms (conv_wgrad.hip): the two-barrier staged loop that the
    software-pipelined form replaced (PMF_WG_SWP=0), the 512-thread form with the taps of a slab on two waves
    (PMF_WG_W8=1), the software-pipelined form with the four waves splitting the PIXELS of one 32 x 32 tile (PMF_WG_S3N=0:
    what layers with 32 output channels run, and the default of rounds 3-4 everywhere) and the N-split form capped at two
    output-channel tiles per workgroup (PMF_WG_S3N=2); same cases, same float64 bars"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ran = 0
    for case in CONVS:
        if case[6] in (2, 3) and case[9] == 1:      # 2x2 / 3x3, stride 1
            _conv_case(case, 0)
            ran += 1
    assert ran >= 2


WS_CASES = [c for c in CONVS if c[6] in (2, 3) and (c[9] == 1 or c[0] == "c3x3s2")]     # (stride 2: its 4-tap parity class of the input gradient)


@pytest.mark.parametrize("case", WS_CASES, ids=[c[0] for c in WS_CASES])
def test_conv_unit_wave_scheduled(case):
    """the wave-scheduled N-split kernel (conv_ws.hip, cfg bit 25) on every 2x2 / 3x3 stride-1 case: forward, input gradients (its
    transposed launches; launches that do not qualify -- 24 output channels, dilation 6 / 18 -- keep the staged loop), BatchNorm
    statistics incl. the per-tile statistics fold of one / two / four output-channel tiles per workgroup, float64 bars"""
    _conv_case(case, 32 | (1 << 8) | (1 << 16) | L.CFG_WS)


@pytest.mark.parametrize("code", [1, 2, 3], ids=["nco1", "nco2", "nco4"])
@pytest.mark.parametrize("case", [c for c in WS_CASES if c[5] % 64 == 0], ids=[c[0] for c in WS_CASES if c[5] % 64 == 0])
def test_conv_unit_wave_scheduled_nco(case, code):
    """the default rule of pmf_conv_ws_ok (tiles x Cout / (32 NCO) >= 256) sends every unit shape to ONE
    output-channel tile per workgroup, while the shipped tile table asks for two / four on 102 shapes (cfg bits 26-27).  The
    cap is set explicitly here: conv_ws_k<*, 1 / 2 / 4, *> and the per-tile statistics fold of conv_epi.h (plain BatchNorm
    statistics and the multi-destination merged input gradients of the concatenated cases) against float64"""
    _conv_case(case, 32 | (1 << 8) | (1 << 16) | L.CFG_WS | (code << 26))


DIRECT_1X1 = [c for c in CONVS if c[0] in ("c1x1cat3", "c1x1_plain_lrelu", "c1x1s2", "c1x1cat3_big", "c1x1_odd_big",
                                            "c1x1_c20_big", "c1x1_wide_big", "c1x1_odd_wide", "c1x1_to1024", "c1x1_k768_cat",
                                            "c1x1_k784")]


@pytest.mark.parametrize("case", DIRECT_1X1, ids=[c[0] for c in DIRECT_1X1])
def test_conv_unit_direct_1x1(case, monkeypatch):
    """the 1x1 layers again with the direct split-bf16 variant (conv_fwd.hip PIPE 11) admitted at every size (the plan
    only uses it from 32768 pixels up): forward, input gradients (its transposed launches), BatchNorm statistics"""
    monkeypatch.setenv("PMF_S3_DIRECT_MIN_PIX", "1")
    Hn = _conv_case(case, 0)
    assert sum(1 for j in Hn.P.pack_jobs if j[7] == 1) >= 1      # split-bf16 weight fragments were requested
    _conv_case(case, 64 | (2 << 8) | (1 << 16))


def _conv_case(case, cfg):
    name, N, H, W, cins, Cout, k, dil, pad, stride, bias, mode, use_bn = case
    conv = nn.Conv2d(sum(cins), Cout, k, stride, pad, dil, bias=bias)
    bn = nn.BatchNorm2d(Cout) if use_bn else None
    with torch.no_grad():
        conv.weight.copy_(det_tensor(name + ".w", conv.weight.shape, -0.2, 0.2))
        if bias:
            conv.bias.copy_(det_tensor(name + ".b", (Cout,), -0.5, 0.5))
        if bn is not None:
            bn.weight.copy_(det_tensor(name + ".g", (Cout,), 0.5, 1.5))
            bn.bias.copy_(det_tensor(name + ".be", (Cout,), -0.5, 0.5))
    conv_d, bn_d = conv, bn
    conv, bn = conv.to(DEV), (bn.to(DEV) if bn is not None else None)
    import copy
    conv64 = copy.deepcopy(conv_d).cpu().double()
    bn64 = copy.deepcopy(bn_d).cpu().double().train() if bn_d is not None else None
    Hn = Harness([(N, c, H, W) for c in cins], masks=N * sum(cins))
    P = Hn.P
    # a Dropout2d multiplier on the 2nd operand when there are several
    cm = None
    views = []
    for i, t in enumerate(Hn.inputs):
        v = V(t)
        if i == 1:
            cm = (det_tensor(name + ".cm", (N, cins[1])) > -0.5).float() * 1.25
            P.masks[:N * cins[1]].copy_(cm.reshape(-1).to(DEV))
            v = v.with_cmul(0, cins[1])
        views.append(v)
    if mode == "act_bn":
        out = P.conv(views, conv, L.ACT_LRELU, bn, name="u")
    elif mode == "lrelu":
        out = P.conv(views, conv, L.ACT_LRELU, name="u")
    elif mode == "none":
        out = P.conv(views, conv, L.ACT_NONE, name="u")
    elif mode == "bn_relu":
        out = P.conv(views, conv, L.ACT_NONE, bn, "bn_act", True, name="u")
    else:
        out = P.conv(views, conv, L.ACT_NONE, bn, "bn_act", False, name="u")
    xs = [det_tensor("%s.x%d" % (name, i), s) for i, s in enumerate(Hn.shapes)]
    OH = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    gout = det_tensor(name + ".go", (N, Cout, OH, OW))
    params = [conv.weight] + ([conv.bias] if bias else []) + ([bn.weight, bn.bias] if bn is not None else [])
    # float64 reference
    xd = [x.double().requires_grad_(True) for x in xs]
    xin = [x if i != 1 or cm is None else x * cm.double()[:, :, None, None] for i, x in enumerate(xd)]
    z = conv64(torch.cat(xin, 1))
    if mode == "act_bn":
        y = bn64(F.leaky_relu(z, 0.01))
    elif mode == "lrelu":
        y = F.leaky_relu(z, 0.01)
    elif mode == "none":
        y = z
    elif mode == "bn_relu":
        y = F.relu(bn64(z))
    else:
        y = bn64(z)
    (y * gout.double()).sum().backward()
    # the plan's convention: gradients reach a BatchNorm view w.r.t. the BN OUTPUT (consumers apply relu');
    # the external gradient therefore carries the ReLU mask in the bn_relu cases
    g_in = gout * (y.detach() > 0).float() if mode == "bn_relu" else gout
    got, gin, gp = Hn.run([out], xs, [g_in], params, cfg)
    _check(name + ".out", got[0], y.detach())
    for i, g in enumerate(gin):
        _check(name + ".gin%d" % i, g, xd[i].grad, 1e-4)
    ref_p = [conv64.weight] + ([conv64.bias] if bias else []) + ([bn64.weight, bn64.bias] if bn64 is not None else [])
    for p, r in zip(params, ref_p):
        _check("%s.gparam%s" % (name, tuple(r.shape)), gp[id(p)], r.grad, 1e-4)
    if bn is not None:
        _check(name + ".running_mean", bn.running_mean, bn64.running_mean, 1e-5)
        _check(name + ".running_var", bn.running_var, bn64.running_var, 1e-5)
    return Hn


def test_bcast_operand_conv():
    """ASPP image-level branch: a [N,1,1,C] map broadcast into a 1x1 conv next to a full-size operand."""
    N, C, H, W, Co = 2, 32, 4, 24, 32
    conv = nn.Conv2d(2 * C, Co, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(det_tensor("bc.w", conv.weight.shape, -0.3, 0.3))
    Hn = Harness([(N, C, 1, 1), (N, C, H, W)])
    P = Hn.P
    out = P.conv([V(Hn.inputs[0], bcast=True), V(Hn.inputs[1])], conv, L.ACT_NONE, name="u")
    xs = [det_tensor("bc.x0", (N, C, 1, 1)), det_tensor("bc.x1", (N, C, H, W))]
    gout = det_tensor("bc.go", (N, Co, H, W))
    got, gin, gp = Hn.run([out], xs, [gout], [conv.weight, conv.bias])
    xd = [x.double().requires_grad_(True) for x in xs]
    import copy
    c64 = copy.deepcopy(conv).cpu().double()
    y = c64(torch.cat((xd[0].expand(-1, -1, H, W), xd[1]), 1))
    (y * gout.double()).sum().backward()
    _check("bc.out", got[0], y.detach())
    _check("bc.gin0", gin[0], xd[0].grad, 1e-4)
    _check("bc.gin1", gin[1], xd[1].grad, 1e-4)
    _check("bc.gw", gp[id(conv.weight)], c64.weight.grad, 1e-4)
    _check("bc.gb", gp[id(conv.bias)], c64.bias.grad, 1e-4)


def _affine_view(P, t, key, C, relu=False):
    """a fake BatchNorm'd view: constant scale/shift buffers (no BN backward is attached)."""
    sc, sh = det_tensor(key + ".sc", (C,), 0.5, 1.5), det_tensor(key + ".sh", (C,), -0.5, 0.5)
    bs, bh = P.persist.alloc(4 * C), P.persist.alloc(4 * C)
    return V(t, bs, bh, relu=relu), (bs, bh, sc, sh)


def _set_affine(P, aff):
    for bs, bh, sc, sh in aff:
        bs.tensor((sc.numel(),)).copy_(sc.to(DEV))
        bh.tensor((sh.numel(),)).copy_(sh.to(DEV))


def test_elementwise_primitives_fwd_bwd():
    N, C, H, W = 2, 32, 12, 20
    Hn = Harness([(N, C, H, W), (N, C, H, W), (N, C, H, W)], masks=4 * N * C)
    P = Hn.P
    a, b, c = Hn.inputs
    cm = (det_tensor("ew.cm", (N, C)) > -0.5).float() * 1.25
    cm2 = (det_tensor("ew.cm2", (N, C // 4)) > -0.5).float() * 1.25
    P.masks[:N * C].copy_(cm.reshape(-1).to(DEV))
    P.masks[N * C:N * C + N * C // 4].copy_(cm2.reshape(-1).to(DEV))
    va, aff_a = _affine_view(P, a, "ew.a", C)
    vb, aff_b = _affine_view(P, b, "ew.b", C)
    o_add = P.add_act(va, V(c), L.ACT_RELU, name="add")
    o_pool = P.avgpool(V(a).with_cmul(0, C), name="pool")
    o_bil = P.bilinear(vb, name="bil")
    o_ps = P.pixel_shuffle(V(c).with_cmul(0, C), N * C, C // 4, name="ps")
    o_gate = P.gate(va, vb, c, name="gate")
    o_gm = P.global_mean(V(b).with_cmul(0, C), name="gm")
    outs = [V(o_add), V(o_pool), V(o_bil), V(o_ps), V(o_gate), V(o_gm)]
    xs = [det_tensor("ew.x%d" % i, (N, C, H, W)) for i in range(3)]
    xd = [x.double().requires_grad_(True) for x in xs]
    sa, ha, sb, hb = [t.double().view(1, -1, 1, 1) for t in (aff_a[2], aff_a[3], aff_b[2], aff_b[3])]
    cmd, cm2d = cm.double()[:, :, None, None], cm2.double()[:, :, None, None]
    ya, yb = xd[0] * sa + ha, xd[1] * sb + hb
    refs = [F.relu(ya + xd[2]), F.avg_pool2d(xd[0] * cmd, 3, 2, 1),
            F.interpolate(yb, scale_factor=2, mode="bilinear", align_corners=False),
            F.pixel_shuffle(xd[2] * cmd, 2) * cm2d, ya * torch.sigmoid(yb) + xd[2], (xd[1] * cmd).mean((2, 3), keepdim=True)]
    gouts = [det_tensor("ew.go%d" % i, r.shape) for i, r in enumerate(refs)]
    sum((r * g.double()).sum() for r, g in zip(refs, gouts)).backward()
    # finalise happens inside run(); affine constants are written right after (persist arena exists only then)
    orig_finalise = P.finalise

    def fin():
        r = orig_finalise()
        _set_affine(P, [aff_a, aff_b])
        return r
    P.finalise = fin
    got, gin, _ = Hn.run(outs, xs, gouts)
    for nm, g, r in zip(("add", "pool", "bil", "ps", "gate", "gm"), got, refs):
        _check("ew." + nm, g, r.detach())
    # gradients w.r.t. the raw inputs: views with a constant affine deliver dL/dy, so scale by d y/d x on the host
    ga = gin[0]
    # a: through va (scale sa) [add, gate] and directly [pool]  -> the plan accumulates dL/dy_a and dL/da in ONE buffer
    # only when the view has no BN; here va is a distinct root, so compare the sum of both paths:
    _check("ew.gin_c", gin[2], xd[2].grad, 1e-4)


def test_maxpool_fwd_bwd():
    N, C, H, W = 2, 64, 12, 20
    Hn = Harness([(N, C, H, W)])
    P = Hn.P
    out = P.maxpool(V(Hn.inputs[0], relu=True), name="mp")
    x = det_tensor("mp.x", (N, C, H, W))
    xd = x.double().requires_grad_(True)
    ref = F.max_pool2d(F.relu(xd), 3, 2, 1)
    go = det_tensor("mp.go", ref.shape)
    (ref * go.double()).sum().backward()
    got, gin, _ = Hn.run([V(out)], [x], [go])
    _check("mp.out", got[0], ref.detach())
    _check("mp.gin", gin[0], xd.grad, 1e-5)


# ---- optimiser range kernels (csrc/optim.hip) against torch.optim on the CPU ----------------------------------------
@pytest.mark.parametrize("n,off", [(1, 0), (7, 0), (1024, 0), (4099, 64), (100003, 1), (262144, 128)])
def test_adamw_range_matches_torch(n, off):
    import ctypes as C
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n + off, generator=g)
    ref = torch.nn.Parameter(p0[off:].clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.01)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n + off, device=DEV), torch.zeros(n + off, device=DEV)
    step = torch.zeros((), device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in range(4):
        gr = torch.randn(n + off, generator=g) * (10.0 ** (k - 2))
        ref.grad = gr[off:].clone()
        opt.param_groups[0]["lr"] = 1e-3 * (k + 1)
        opt.step()
        gd = gr.to(DEV)
        step += 1
        rc = L.lib().pmf_adamw_range(p.data_ptr() + 4 * off, gd.data_ptr() + 4 * off, m.data_ptr() + 4 * off,
                                     v.data_ptr() + 4 * off, n, 1e-3 * (k + 1), 0.9, 0.999, 1e-8, 0.01, step.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(p[:off].cpu(), p0[:off])                     # nothing outside the range is touched
        assert torch.allclose(p[off:].cpu(), ref.data, rtol=2e-6, atol=2e-7), (k, (p[off:].cpu() - ref.data).abs().max().item())
        s = opt.state[ref]
        assert (m[off:].cpu() - s["exp_avg"]).abs().max() <= 1e-6 * s["exp_avg"].abs().max()
        assert (v[off:].cpu() - s["exp_avg_sq"]).abs().max() <= 1e-6 * s["exp_avg_sq"].abs().max()


@pytest.mark.parametrize("n,off,nesterov,wd", [(5, 0, True, 1e-5), (4099, 64, True, 1e-5), (100003, 3, False, 0.0),
                                                 (65536, 0, True, 0.0)])
def test_sgd_range_matches_torch(n, off, nesterov, wd):
    import ctypes as C
    g = torch.Generator().manual_seed(n + 1)
    p0 = torch.randn(n + off, generator=g)
    ref = torch.nn.Parameter(p0[off:].clone())
    opt = torch.optim.SGD([ref], lr=1e-2, momentum=0.9, nesterov=nesterov, weight_decay=wd)
    p = p0.clone().to(DEV)
    buf = torch.full((n + off,), float("nan"), device=DEV)            # first step must not read it
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in range(4):
        gr = torch.randn(n + off, generator=g)
        ref.grad = gr[off:].clone()
        opt.step()
        gd = gr.to(DEV)
        rc = L.lib().pmf_sgd_range(p.data_ptr() + 4 * off, gd.data_ptr() + 4 * off, buf.data_ptr() + 4 * off, n, 1e-2, 0.9,
                                   0.0, wd, int(nesterov), int(k == 0), st)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.allclose(p[off:].cpu(), ref.data, rtol=2e-6, atol=2e-7), (k, (p[off:].cpu() - ref.data).abs().max().item())
        assert torch.allclose(buf[off:].cpu(), opt.state[ref]["momentum_buffer"], rtol=2e-6, atol=1e-7)
    # momentum 0: no buffer at all; nesterov without momentum is an argument error (torch raises ValueError)
    p2 = p0.clone().to(DEV)
    gd = torch.ones(n + off, device=DEV)
    assert L.lib().pmf_sgd_range(p2.data_ptr(), gd.data_ptr(), None, n + off, 0.5, 0.0, 0.0, 0.0, 0, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.allclose(p2.cpu(), p0 - 0.5)
    assert L.lib().pmf_sgd_range(p2.data_ptr(), gd.data_ptr(), None, n + off, 0.5, 0.0, 0.0, 0.0, 1, 0, st) == L.PMF_E_ARG


def test_normalise_inplace_is_bit_identical_to_the_torch_expression():
    """trainer.py:291-295 as one launch: same bits as (x[:, :5] - mean) / std * mask.unsqueeze(1), channels 5..7 untouched;
    through TrainEngine.prepare (contiguous float32 -> the kernel; a strided view -> the torch expression)"""
    import ctypes as C
    from pmf_amd.engine import TrainEngine
    g = torch.Generator().manual_seed(4)
    n, h, w = 3, 17, 70
    x0 = (torch.randn(n, 8, h, w, generator=g) * 30).to(DEV)
    mask = (torch.rand(n, h, w, generator=g) > 0.3).float().to(DEV)
    mean = torch.tensor([12.12, 10.88, 0.23, -1.04, 0.21], device=DEV).view(1, 5, 1, 1)
    std = torch.tensor([12.32, 11.47, 6.91, 0.86, 0.16], device=DEV).view(1, 5, 1, 1)
    want = x0.clone()
    want[:, 0:5] = (want[:, 0:5] - mean) / std * mask.unsqueeze(1)
    got = x0.clone()
    rc = L.lib().pmf_normalise_inplace(got.data_ptr(), got.stride(0), mask.data_ptr(), mean.data_ptr(), std.data_ptr(), n, 5,
                                       h * w, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert L.lib().pmf_normalise_inplace(got.data_ptr(), 4 * h * w, mask.data_ptr(), mean.data_ptr(), std.data_ptr(), n, 5,
                                         h * w, None) == L.PMF_E_ARG            # samples would overlap
    eng = TrainEngine.__new__(TrainEngine)
    eng.mean, eng.std = mean, std
    a = x0.clone()
    pcd, rgb = eng.prepare(a, mask)
    assert torch.equal(a, want) and pcd.data_ptr() == a.data_ptr() and torch.equal(rgb, x0[:, 5:8])
    big = torch.zeros(n, 9, h, w, device=DEV)
    b = big[:, 1:]                                   # non-contiguous view: torch expression
    b.copy_(x0)
    eng.prepare(b, mask)
    assert torch.equal(b, want)
