"""PMFNet / SalsaNext on MI355X: the reference's module tree as PARAMETER CONTAINERS + a static HIP plan.

API surface mirrored from the reference:
    PMFNet(pcd_channels=5, img_channels=3, nclasses=20, base_channels=32, imagenet_pretrained=True,
           image_backbone="resnet34")                                    pc_processor/models/pmf_net.py:224-249
    model(pcd[N,5,H,W], rgb[N,3,H,W]) -> (lidar_prob, camera_prob), each [N,nclasses,H,W] softmaxed
    .lidar_stream / .camera_stream_encoder / .camera_stream_decoder; state_dict keys identical (654 for R34).

The sub-modules (nn.Conv2d / nn.BatchNorm2d / nn.Sequential) only HOLD parameters and buffers -- their own
``forward`` is never called.  ``PMFNet.forward`` runs libpmf_amd.so through a static plan (pmf_amd/plan.py);
there is no PyTorch / CPU fallback: on a CPU tensor or without the shared library it raises.
"""
import warnings

import os

import torch
import torch.nn as nn

from .. import _lib as L
from ..plan import Plan, V

__all__ = ["PMFNet", "SalsaNext", "SalsaNextFusion", "ResNet", "RGBDecoder", "ASPP",
           "ResidualBasedFusionBlock", "ResContextBlock", "ResBlock", "UpBlock"]


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s only holds parameters; call the top-level model (HIP plan) instead"
                           % type(self).__name__)


# ------------------------------------------------------------------------------------------------------
# SalsaNext blocks (pc_processor/models/salsanext.py)
# ------------------------------------------------------------------------------------------------------
class ResContextBlock(_Holder):
    """salsanext.py:9-36."""

    def __init__(self, in_filters, out_filters):
        super().__init__()
        self.conv1 = nn.Conv2d(in_filters, out_filters, (1, 1))
        self.act1 = nn.LeakyReLU()
        self.conv2 = nn.Conv2d(out_filters, out_filters, (3, 3), padding=1)
        self.act2 = nn.LeakyReLU()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, (3, 3), dilation=2, padding=2)
        self.act3 = nn.LeakyReLU()
        self.bn2 = nn.BatchNorm2d(out_filters)

    def emit(self, P, x, name):
        s = P.conv([x], self.conv1, L.ACT_LRELU, name=name + ".s")
        a1 = P.conv([s], self.conv2, L.ACT_LRELU, self.bn1, name=name + ".a1")
        a2 = P.conv([a1], self.conv3, L.ACT_LRELU, self.bn2, name=name + ".a2")
        return V(P.add_act(s, a2, L.ACT_NONE, name=name + ".out"))


class ResBlock(_Holder):
    """salsanext.py:38-104."""

    def __init__(self, in_filters, out_filters, dropout_rate, kernel_size=(3, 3), stride=1, pooling=True,
                 drop_out=True):
        super().__init__()
        self.pooling, self.drop_out = pooling, drop_out
        self.conv1 = nn.Conv2d(in_filters, out_filters, (1, 1), stride=stride)
        self.act1 = nn.LeakyReLU()
        self.conv2 = nn.Conv2d(in_filters, out_filters, (3, 3), padding=1)
        self.act2 = nn.LeakyReLU()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, (3, 3), dilation=2, padding=2)
        self.act3 = nn.LeakyReLU()
        self.bn2 = nn.BatchNorm2d(out_filters)
        self.conv4 = nn.Conv2d(out_filters, out_filters, (2, 2), dilation=2, padding=1)
        self.act4 = nn.LeakyReLU()
        self.bn3 = nn.BatchNorm2d(out_filters)
        self.conv5 = nn.Conv2d(out_filters * 3, out_filters, (1, 1))
        self.act5 = nn.LeakyReLU()
        self.bn4 = nn.BatchNorm2d(out_filters)
        self.dropout = nn.Dropout2d(p=dropout_rate)
        if pooling:
            self.pool = nn.AvgPool2d(kernel_size=kernel_size, stride=2, padding=1)

    def emit(self, P, x, name, drop=None):
        """returns (pooled T, skip T) when pooling else the (dropped) view."""
        s = P.conv([x], self.conv1, L.ACT_LRELU, name=name + ".s")
        r1 = P.conv([x], self.conv2, L.ACT_LRELU, self.bn1, name=name + ".r1")
        r2 = P.conv([r1], self.conv3, L.ACT_LRELU, self.bn2, name=name + ".r2")
        r3 = P.conv([r2], self.conv4, L.ACT_LRELU, self.bn3, name=name + ".r3")
        r5 = P.conv([r1, r2, r3], self.conv5, L.ACT_LRELU, self.bn4, name=name + ".r5")
        res_a = P.add_act(s, r5, L.ACT_NONE, name=name + ".resA")
        vb = V(res_a)
        if self.drop_out and drop is not None:
            vb = vb.with_cmul(drop, res_a.C)
        if self.pooling:
            return P.avgpool(vb, name=name + ".pool"), res_a
        return vb


class UpBlock(_Holder):
    """salsanext.py:107-164."""

    def __init__(self, in_filters, out_filters, dropout_rate, drop_out=True):
        super().__init__()
        self.drop_out, self.in_filters, self.out_filters = drop_out, in_filters, out_filters
        self.dropout1 = nn.Dropout2d(p=dropout_rate)
        self.dropout2 = nn.Dropout2d(p=dropout_rate)
        self.conv1 = nn.Conv2d(in_filters // 4 + 2 * out_filters, out_filters, (3, 3), padding=1)
        self.act1 = nn.LeakyReLU()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv2 = nn.Conv2d(out_filters, out_filters, (3, 3), dilation=2, padding=2)
        self.act2 = nn.LeakyReLU()
        self.bn2 = nn.BatchNorm2d(out_filters)
        self.conv3 = nn.Conv2d(out_filters, out_filters, (2, 2), dilation=2, padding=1)
        self.act3 = nn.LeakyReLU()
        self.bn3 = nn.BatchNorm2d(out_filters)
        self.conv4 = nn.Conv2d(out_filters * 3, out_filters, (1, 1))
        self.act4 = nn.LeakyReLU()
        self.bn4 = nn.BatchNorm2d(out_filters)
        self.dropout3 = nn.Dropout2d(p=dropout_rate)

    def emit(self, P, x, skip, name, masks=None):
        """masks: dict(comb=off, d2=off, d3=off) float offsets into the plan's multiplier tensor, or None."""
        c4 = self.in_filters // 4
        ld2 = c4 + 2 * self.out_filters
        use = self.drop_out and masks is not None
        up_a = P.pixel_shuffle(x, masks["comb"] if use else None, c4, name=name + ".upA")
        sk = V(skip)
        if use:
            sk = sk.with_cmul(masks["d2"] + c4, ld2)
        e1 = P.conv([V(up_a), sk], self.conv1, L.ACT_LRELU, self.bn1, name=name + ".e1")
        e2 = P.conv([e1], self.conv2, L.ACT_LRELU, self.bn2, name=name + ".e2")
        e3 = P.conv([e2], self.conv3, L.ACT_LRELU, self.bn3, name=name + ".e3")
        e = P.conv([e1, e2, e3], self.conv4, L.ACT_LRELU, self.bn4, name=name + ".e")
        if use:
            e = e.with_cmul(masks["d3"], self.out_filters)
        return e


class ResidualBasedFusionBlock(_Holder):
    """pmf_net.py:10-36."""

    def __init__(self, pcd_channels, img_channels):
        super().__init__()
        p = pcd_channels
        self.fuse_conv = nn.Sequential(nn.Conv2d(p + img_channels, p, 3, padding=1, stride=1), nn.LeakyReLU(),
                                       nn.BatchNorm2d(p))
        self.attention = nn.Sequential(nn.Conv2d(p, p, 3, padding=1, stride=1), nn.BatchNorm2d(p),
                                       nn.ReLU(inplace=True), nn.Conv2d(p, p, 3, padding=1, stride=1),
                                       nn.BatchNorm2d(p), nn.Sigmoid())

    def emit(self, P, pcd, img, name):
        f = P.conv([V(pcd), img], self.fuse_conv[0], L.ACT_LRELU, self.fuse_conv[2], name=name + ".f")
        a1 = P.conv([f], self.attention[0], L.ACT_NONE, self.attention[1], "bn_act", True, name=name + ".a1")
        a2 = P.conv([a1], self.attention[3], L.ACT_NONE, self.attention[4], "bn_act", False, name=name + ".a2")
        return P.gate(f, a2, pcd, name=name + ".out")


class ASPP(_Holder):
    """pmf_net.py:103-138 (no BN, no activation)."""

    def __init__(self, in_channel=512, depth=256):
        super().__init__()
        self.mean = nn.AdaptiveAvgPool2d((1, 1))
        self.conv = nn.Conv2d(in_channel, depth, 1, 1)
        self.atrous_block1 = nn.Conv2d(in_channel, depth, 1, 1)
        self.atrous_block6 = nn.Conv2d(in_channel, depth, 3, 1, padding=6, dilation=6)
        self.atrous_block12 = nn.Conv2d(in_channel, depth, 3, 1, padding=12, dilation=12)
        self.atrous_block18 = nn.Conv2d(in_channel, depth, 3, 1, padding=18, dilation=18)
        self.conv_1x1_output = nn.Conv2d(depth * 5, depth, 1, 1)

    def emit(self, P, x, name):
        gm = P.global_mean(x, name=name + ".mean")
        g = P.conv([V(gm)], self.conv, name=name + ".imgfeat")
        # the image-level branch as a MATERIALISED operand (round 6; a broadcast-flagged operand, V(g.t, bcast=True), keeps the
        # projection below on the generic loop and its input gradient in five launches)
        gb = V(P.broadcast(g.t, x.t.H, x.t.W, name=name + ".imgfeat_b"))
        b1 = P.conv([x], self.atrous_block1, name=name + ".b1")
        b6 = P.conv([x], self.atrous_block6, name=name + ".b6")
        b12 = P.conv([x], self.atrous_block12, name=name + ".b12")
        b18 = P.conv([x], self.atrous_block18, name=name + ".b18")
        return P.conv([gb, b1, b6, b12, b18], self.conv_1x1_output, name=name + ".out")


class SalsaNext(_Holder):
    """salsanext.py:166-208.  Stand-alone LiDAR-only model (also the base of the fusion trunk)."""

    def __init__(self, in_channels=8, nclasses=20, base_channels=32, softmax=True):
        super().__init__()
        c = base_channels
        self.base_channels, self.dropout_ratio, self.softmax = c, 0.2, softmax
        self.in_channels, self.nclasses = in_channels, nclasses
        self.downCntx = ResContextBlock(in_channels, c)
        self.downCntx2 = ResContextBlock(c, c)
        self.downCntx3 = ResContextBlock(c, c)
        self.resBlock1 = ResBlock(c, 2 * c, 0.2, pooling=True, drop_out=False)
        self.resBlock2 = ResBlock(2 * c, 4 * c, 0.2, pooling=True)
        self.resBlock3 = ResBlock(4 * c, 8 * c, 0.2, pooling=True)
        self.resBlock4 = ResBlock(8 * c, 8 * c, 0.2, pooling=True)
        self.resBlock5 = ResBlock(8 * c, 8 * c, 0.2, pooling=False)
        self.upBlock1 = UpBlock(8 * c, 4 * c, 0.2)
        self.upBlock2 = UpBlock(4 * c, 4 * c, 0.2)
        self.upBlock3 = UpBlock(4 * c, 2 * c, 0.2)
        self.upBlock4 = UpBlock(2 * c, c, 0.2, drop_out=False)
        self.logits = nn.Conv2d(c, nclasses, kernel_size=(1, 1))
        self._plans = {}

    # hooks overridden by the fusion trunk
    def _fuse(self, P, i, x, feats):
        return x

    def _bottleneck(self, P, x):
        return x

    def emit_trunk(self, P, x, feats, M):
        """x: V of the [N,H,W,8] LiDAR input; feats: camera features (or None); M: mask offsets or None."""
        m = (lambda k: M[k]) if M is not None else (lambda k: None)
        d = self.downCntx.emit(P, x, "downCntx")
        d = self.downCntx2.emit(P, d, "downCntx2")
        d = self.downCntx3.emit(P, d, "downCntx3")
        d0c, d0b = self.resBlock1.emit(P, d, "resBlock1")
        d0c = self._fuse(P, 1, d0c, feats)
        d1c, d1b = self.resBlock2.emit(P, V(d0c), "resBlock2", m("resBlock2"))
        d1c = self._fuse(P, 2, d1c, feats)
        d2c, d2b = self.resBlock3.emit(P, V(d1c), "resBlock3", m("resBlock3"))
        d2c = self._fuse(P, 3, d2c, feats)
        d3c, d3b = self.resBlock4.emit(P, V(d2c), "resBlock4", m("resBlock4"))
        d3c = self._fuse(P, 4, d3c, feats)
        d5c = self._bottleneck(P, self.resBlock5.emit(P, V(d3c), "resBlock5", m("resBlock5")))
        um = (lambda i: dict(comb=M["upBlock%d.comb" % i], d2=M["upBlock%d.d2" % i], d3=M["upBlock%d.d3" % i])) \
            if M is not None else (lambda i: None)
        u = self.upBlock1.emit(P, d5c, d3b, "upBlock1", um(1))
        u = self.upBlock2.emit(P, u, d2b, "upBlock2", um(2))
        u = self.upBlock3.emit(P, u, d1b, "upBlock3", um(3))
        u = self.upBlock4.emit(P, u, d0b, "upBlock4", None)
        lg = P.conv([u], self.logits, name="logits")
        P.softmax_out(lg.t, "lidar", "lidar", softmax=self.softmax)
        return lg

    # ---- stand-alone use -------------------------------------------------------------------------
    def forward(self, x):
        return _run_model(self, (x,))[0]

    def _build(self, N, H, W, training, device, dry=False):
        P = Plan(device, training, getattr(self, "_flat", None), dry)
        M = _alloc_masks(P, self, N, device) if training else None
        x = V(P.input_nchw("pcd", N, self.in_channels, H, W, "pcd"))
        self.emit_trunk(P, x, None, M)
        return P.finalise()

    def _mask_sites(self):
        return _salsa_sites(self)

    def _apply(self, fn, *a, **k):
        self._plans = {}
        self._flat = None          # storage is about to move: a FlatState must be re-created afterwards
        return super()._apply(fn, *a, **k)


class SalsaNextFusion(SalsaNext):
    """pmf_net.py:141-180."""

    def __init__(self, in_channels=8, nclasses=20, base_channels=32, img_feature_channels=[]):
        super().__init__(in_channels=in_channels, base_channels=base_channels, nclasses=nclasses, softmax=True)
        c = self.base_channels
        self.fusionblock_1 = ResidualBasedFusionBlock(c * 2, img_feature_channels[0])
        self.fusionblock_2 = ResidualBasedFusionBlock(c * 4, img_feature_channels[1])
        self.fusionblock_3 = ResidualBasedFusionBlock(c * 8, img_feature_channels[2])
        self.fusionblock_4 = ResidualBasedFusionBlock(c * 8, img_feature_channels[3])
        self.aspp = ASPP(c * 8, c * 8)

    def _fuse(self, P, i, x, feats):
        ahead = getattr(P, "enc_ahead", 0)   # encoder stages emitted this many fusion blocks before their first reader
        if ahead:
            feats[min(3, i - 1 + ahead)]
        f = feats[i - 1]                     # (a lazy sequence emits the encoder stage here, on its own lane)
        ready = getattr(P, "feat_ready", None)
        if ready:                            # the camera features come from another lane
            P.wait_event(P.fwd, ready[i - 1])
        return getattr(self, "fusionblock_%d" % i).emit(P, x, f, "fusion%d" % i)

    def _bottleneck(self, P, x):
        return self.aspp.emit(P, x, "aspp")

    def forward(self, *a, **k):
        raise RuntimeError("SalsaNextFusion runs inside PMFNet's plan; call PMFNet")


# ------------------------------------------------------------------------------------------------------
# camera stream: torchvision-structured ResNet (third-party arithmetic, restated) + decoder
# ------------------------------------------------------------------------------------------------------
class BasicBlock(_Holder):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def emit(self, P, x, name):
        y = P.conv([x], self.conv1, L.ACT_NONE, self.bn1, "bn_act", True, name=name + ".c1")
        y = P.conv([y], self.conv2, L.ACT_NONE, self.bn2, "bn_act", False, name=name + ".c2")
        idt = x
        if self.downsample is not None:
            idt = P.conv([x], self.downsample[0], L.ACT_NONE, self.downsample[1], "bn_act", False, name=name + ".ds")
        return V(P.add_act(y, idt, L.ACT_RELU, name=name + ".out"))


class Bottleneck(_Holder):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def emit(self, P, x, name):
        y = P.conv([x], self.conv1, L.ACT_NONE, self.bn1, "bn_act", True, name=name + ".c1")
        y = P.conv([y], self.conv2, L.ACT_NONE, self.bn2, "bn_act", True, name=name + ".c2")
        y = P.conv([y], self.conv3, L.ACT_NONE, self.bn3, "bn_act", False, name=name + ".c3")
        idt = x
        if self.downsample is not None:
            idt = P.conv([x], self.downsample[0], L.ACT_NONE, self.downsample[1], "bn_act", False, name=name + ".ds")
        return V(P.add_act(y, idt, L.ACT_RELU, name=name + ".out"))


_RESNET_CFG = {"resnet34": (BasicBlock, (3, 4, 6, 3)), "resnet50": (Bottleneck, (3, 4, 6, 3)),
               "resnet101": (Bottleneck, (3, 4, 23, 3)), "resnet152": (Bottleneck, (3, 8, 36, 3))}


def _make_layer(block, inplanes, planes, blocks, stride):
    ds = None
    if stride != 1 or inplanes != planes * block.expansion:
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                           nn.BatchNorm2d(planes * block.expansion))
    layers = [block(inplanes, planes, stride, ds)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class ResNet(_Holder):
    """pmf_net.py:41-100: torchvision resnet{34,50,101,152} body, stride-1 7x7 stem, Dropout2d on layer3/4."""

    def __init__(self, in_channels=3, backbone="resnet50", dropout_rate=0.2, pretrained=True):
        super().__init__()
        if backbone not in _RESNET_CFG:
            raise NotImplementedError("invalid backbone: {}".format(backbone))
        block, counts = _RESNET_CFG[backbone]
        self.expansion = block.expansion
        e = self.expansion
        self.feature_channels = [64 * e, 128 * e, 256 * e, 512 * e]
        self.backbone_name = backbone
        self.in_channels = in_channels
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = _make_layer(block, 64, 64, counts[0], 1)
        self.layer2 = _make_layer(block, 64 * e, 128, counts[1], 2)
        self.layer3 = _make_layer(block, 128 * e, 256, counts[2], 2)
        self.layer4 = _make_layer(block, 256 * e, 512, counts[3], 2)
        self.dropout = nn.Dropout2d(p=dropout_rate)
        for m in self.modules():   # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if pretrained:
            _try_load_imagenet(self, backbone, in_channels)

    def emit(self, P, x, M, ready=None):
        """ready (optional list): receives one plan event per feature map, recorded on the emitting lane once the
        feature is complete -- consumers on another lane wait for it (Plan.wait_event)."""
        feats = self.emit_lazy(P, x, M, ready, P.lane)
        feats[3]
        return list(feats.done)

    def emit_lazy(self, P, x, M, ready, lane):
        """the four feature maps as a sequence that emits a stage (on ``lane``) the first time it is indexed: the fusion
        trunk asks for feature i right where it reads it, so encoder stage i sits in the op list next to its consumer
        instead of in front of the whole trunk.  Forward execution is the same (lanes are ordered by events, not by list
        position); the BACKWARD list -- the reverse -- then has every encoder stage right behind the fusion block that
        feeds its gradient, which is what keeps both lanes busy when the backward plan runs in segments (data-parallel
        path): with the whole encoder at the end of the list the last segment was the camera stream alone (+1.8 ms)."""
        enc = self

        class Lazy:
            def __init__(self):
                self.done, self.y = [], None

            def __getitem__(self, k):
                while len(self.done) <= k:
                    li = len(self.done)
                    keep, P.lane = P.lane, lane
                    if li == 0:
                        c1 = P.conv([x], enc.conv1, L.ACT_NONE, enc.bn1, "bn_act", True, name="enc.stem")
                        self.y = V(P.maxpool(c1, name="enc.maxpool"))
                    y = self.y
                    for bi, blk in enumerate((enc.layer1, enc.layer2, enc.layer3, enc.layer4)[li]):
                        y = blk.emit(P, y, "enc.layer%d.%d" % (li + 1, bi))
                    if li >= 2 and M is not None:    # Dropout2d on layer3 / layer4 outputs (one module, two masks)
                        y = y.with_cmul(M["enc.f%d" % li], y.t.C)
                    self.y = y
                    self.done.append(y)
                    if ready is not None:
                        ready.append(P.record_event(P.fwd))
                    P.lane = keep
                return self.done[k]
        return Lazy()


def _try_load_imagenet(net, backbone, in_channels):
    try:
        import torchvision.models.resnet as tvr   # noqa: WPS433 (optional dependency)
        ref = getattr(tvr, backbone)(True)
        sd = {k: v for k, v in ref.state_dict().items() if not k.startswith("fc.")}
        if in_channels != 3:
            sd.pop("conv1.weight")
        net.load_state_dict(sd, strict=False)
    except Exception as e:  # no torchvision / no network: keep the random init, say so
        warnings.warn("imagenet_pretrained=True but ImageNet weights are unavailable (%s); "
                      "camera backbone keeps its random initialisation" % type(e).__name__)


class RGBDecoder(_Holder):
    """pmf_net.py:183-222."""

    def __init__(self, in_channels=[], nclasses=4, base_channels=64):
        super().__init__()
        b = base_channels

        def up(cin, k):
            return nn.Sequential(nn.Conv2d(cin, b, k, padding=k // 2), nn.LeakyReLU(), nn.BatchNorm2d(b),
                                 nn.Upsample(scale_factor=2, mode="bilinear"))
        self.up_4a = up(in_channels[3], 3)
        self.up_3a = up(in_channels[2] + b, 3)
        self.up_2a = up(in_channels[1] + b, 3)
        self.up_1a = up(in_channels[0] + b, 1)
        self.conv = nn.Conv2d(b, nclasses, kernel_size=3, padding=1)

    def emit(self, P, feats):
        def up(seq, srcs, name):
            v = P.conv(srcs, seq[0], L.ACT_LRELU, seq[2], name=name)
            return V(P.bilinear(v, name=name + ".up"))
        u = up(self.up_4a, [feats[3]], "dec.up4")
        u = up(self.up_3a, [u, feats[2]], "dec.up3")
        u = up(self.up_2a, [u, feats[1]], "dec.up2")
        u = up(self.up_1a, [u, feats[0]], "dec.up1")
        lg = P.conv([u], self.conv, name="dec.logits")
        P.softmax_out(lg.t, "camera", "camera")
        return lg


class PMFNet(nn.Module):
    """pmf_net.py:224-249 -- dual-branch fusion network, executed as one static HIP plan per input shape."""

    def __init__(self, pcd_channels=5, img_channels=3, nclasses=20, base_channels=32, imagenet_pretrained=True,
                 image_backbone="resnet34"):
        super().__init__()
        self.camera_stream_encoder = ResNet(in_channels=img_channels, pretrained=imagenet_pretrained,
                                            backbone=image_backbone)
        self.camera_stream_decoder = RGBDecoder(self.camera_stream_encoder.feature_channels, nclasses=nclasses,
                                                base_channels=self.camera_stream_encoder.expansion * 16)
        self.lidar_stream = SalsaNextFusion(in_channels=pcd_channels, nclasses=nclasses,
                                            base_channels=base_channels,
                                            img_feature_channels=self.camera_stream_encoder.feature_channels)
        self.pcd_channels, self.img_channels, self.nclasses = pcd_channels, img_channels, nclasses
        self._plans = {}

    def forward(self, pcd_feature, img_feature):
        h, w = img_feature.shape[2], img_feature.shape[3]
        if h % 16 != 0 or w % 16 != 0:   # pmf_net.py:85-88
            assert False, "invalid input size: {}".format(img_feature.shape)
        if pcd_feature.shape[1] != self.pcd_channels or img_feature.shape[1] != self.img_channels or \
                pcd_feature.shape[0] != img_feature.shape[0] or pcd_feature.shape[2:] != img_feature.shape[2:]:
            raise ValueError("PMFNet: expected [N,%d,H,W] and [N,%d,H,W] inputs, got %s and %s" % (
                self.pcd_channels, self.img_channels, tuple(pcd_feature.shape), tuple(img_feature.shape)))
        return _run_model(self, (pcd_feature, img_feature))

    def _build(self, N, H, W, training, device, dry=False):
        P = Plan(device, training, getattr(self, "_flat", None), dry)
        M = _alloc_masks(P, self, N, device) if training else None
        # two lanes (csrc/plan.cpp): the camera stream (encoder + decoder) runs on lane 1, the LiDAR stream on lane 0; the
        # only forward edges between them are the four feature maps the fusion blocks read (pmf_net.py:160-176)
        pcd = V(P.input_nchw("pcd", N, self.pcd_channels, H, W, "pcd"))
        P.lane = 1
        rgb = V(P.input_nchw("rgb", N, self.img_channels, H, W, "rgb"))
        P.feat_ready = []
        # interleaved with the trunk when the backward plan will run in segments (data-parallel engine: its hook is set
        # before the first forward); as one range the plain order measures 0.9 ms faster (11.6 vs 12.5 ms backward), in
        # four segments the interleaved one (12.45 vs 13.4 ms)
        lazy = os.environ.get("PMF_ENC_LAZY")
        lazy = (getattr(self, "_bwd_segment_hook", None) is not None) if lazy is None else lazy != "0"
        # ... one fusion block EARLY: its backward then sits one block later in the list, which measured best in segments
        # (12.17 vs 12.54 ms; as one range 11.65 = the plain order)
        P.enc_ahead = 1
        if lazy:
            feats = self.camera_stream_encoder.emit_lazy(P, rgb, M, P.feat_ready, 1)
        else:
            feats = self.camera_stream_encoder.emit(P, rgb, M, P.feat_ready)
        # (the position of an op in the list is also its dispatch priority inside the replayed graph: the decoder in front
        # of the trunk costs 7 % of the step, the main lane's whole backward in front of the side lane's 20 % of it)
        P.lane = 0
        self.lidar_stream.emit_trunk(P, pcd, feats, M)
        P.lane = 1
        feats[3]
        self.camera_stream_decoder.emit(P, feats)
        P.lane = 0
        return P.finalise()

    def _mask_sites(self):
        enc = self.camera_stream_encoder
        return [("enc.f2", enc.feature_channels[2]), ("enc.f3", enc.feature_channels[3])] + \
            _salsa_sites(self.lidar_stream)

    def _apply(self, fn, *a, **k):
        self._plans = {}
        self._flat = None          # storage is about to move: a FlatState must be re-created afterwards
        return super()._apply(fn, *a, **k)

    # test / reproducibility hook: masks = {site: [N, C] tensor of 0 or 1/(1-p)}; None -> draw from torch RNG
    def set_dropout_masks(self, masks):
        self._forced_masks = masks


# ------------------------------------------------------------------------------------------------------
# Dropout2d multipliers: one float tensor per plan, [site0 | site1 | ... | derived], each [N, C] row-major
# ------------------------------------------------------------------------------------------------------
def _salsa_sites(ls):
    c = ls.base_channels
    sites = [("resBlock%d" % i, ch) for i, ch in ((2, 4 * c), (3, 8 * c), (4, 8 * c), (5, 8 * c))]
    for i, (cin, cout) in ((1, (8 * c, 4 * c)), (2, (4 * c, 4 * c)), (3, (4 * c, 2 * c))):
        sites += [("upBlock%d.d1" % i, cin // 4), ("upBlock%d.d2" % i, cin // 4 + 2 * cout),
                  ("upBlock%d.d3" % i, cout)]
    return sites


def _alloc_masks(P, model, N, device):
    sites = model._mask_sites()
    off, table = 0, {}
    for name, ch in sites:
        table[name] = off
        off += N * ch
    n_sites = off
    derived = []
    for i in (1, 2, 3):
        c4 = dict(sites)["upBlock%d.d1" % i]
        table["upBlock%d.comb" % i] = off
        derived.append((off, table["upBlock%d.d1" % i], table["upBlock%d.d2" % i], c4,
                        dict(sites)["upBlock%d.d2" % i]))
        off += N * c4
    P.masks = torch.ones(max(off, 4), dtype=torch.float32, device=device)
    P.mask_table, P.mask_sites, P.mask_n_sites, P.mask_derived, P.mask_N = table, sites, n_sites, derived, N
    return table


def _fill_masks(P, model, p=0.2):
    forced = getattr(model, "_forced_masks", None)
    N = P.mask_N
    if forced is not None:
        for name, ch in P.mask_sites:
            o = P.mask_table[name]
            P.masks[o:o + N * ch].copy_(forced[name].reshape(-1).to(P.masks))
    else:
        P.masks[:P.mask_n_sites].bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
    for (o, o1, o2, c4, ld2) in P.mask_derived:
        d1 = P.masks[o1:o1 + N * c4].view(N, c4)
        d2 = P.masks[o2:o2 + N * ld2].view(N, ld2)[:, :c4]
        torch.mul(d1, d2, out=P.masks[o:o + N * c4].view(N, c4))


# ------------------------------------------------------------------------------------------------------
# flat training state: parameters and gradients of the whole network in two contiguous buffers
# ------------------------------------------------------------------------------------------------------
class FlatState:
    """All trainable parameters re-homed as views of ONE flat buffer (``param``) with their gradients as views of a
    second one (``grad``), group by group (e.g. [LiDAR stream | camera stream]) and, inside a group, in the order the
    backward plan produces the gradients.  What it buys:
      * the backward plan writes weight gradients straight into ``p.grad`` -- no flat copy, no 654 views per step;
      * an optimiser step is one fused launch per group over one tensor (AdamW / SGD over 654 tensors cost ~1.5 ms of
        host time per iteration during which the GPU idled);
      * a data-parallel all-reduce is a handful of large contiguous collectives whose ranges complete front to back
        while the backward plan is still running (ranges[g] = (start, end) floats; done_after[op index] frontier)."""

    def __init__(self, model, groups, device):
        plan = model._build(1, 64, 64, True, device, dry=True)     # records the backward emission order only
        order = list(plan.params)
        seen = {id(p) for p in order}
        for p in model.parameters():
            if p.requires_grad and id(p) not in seen:
                order.append(p)
                seen.add(id(p))
        gid = {}
        for g, ps in enumerate(groups):
            for p in ps:
                gid[id(p)] = g
        self.offset, self.ranges, self.members = {}, [], []
        # writes to ``param`` through raw pointers (the range optimiser kernels) -- torch's version counter does not see
        # them; (param._version, raw_writes) is what a plan remembers its packed weights by (stamp())
        self.raw_writes = 0
        off = 0
        for g in range(len(groups)):
            start = off
            mem = [p for p in order if gid.get(id(p)) == g and p.requires_grad]
            for p in mem:
                self.offset[id(p)] = off
                off += (p.numel() + 63) // 64 * 64
            self.ranges.append((start, off))
            self.members.append(mem)
        missing = [p for p in order if id(p) not in self.offset]
        if missing:
            raise ValueError("FlatState: %d trainable parameters belong to no group" % len(missing))
        self.param = torch.zeros(max(off, 64), dtype=torch.float32, device=device)
        self.grad = torch.zeros(max(off, 64), dtype=torch.float32, device=device)
        with torch.no_grad():
            for mem in self.members:
                for p in mem:
                    o = self.offset[id(p)]
                    v = self.param[o:o + p.numel()].view(p.shape)
                    v.copy_(p.data)
                    p.data = v
                    p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.group_params = []
        for (a, b) in self.ranges:       # one leaf tensor per group for the optimisers
            fp = torch.nn.Parameter(self.param[a:b], requires_grad=True)
            fp.grad = self.grad[a:b]
            self.group_params.append(fp)

    def stamp(self):
        """changes whenever a parameter is written: by torch through the flat buffer or a group leaf (their shared version
        counter), through a module parameter (each keeps a counter of its own: ``p.data = view`` does not share it), or by
        the range optimiser kernels, which write through raw pointers (raw_writes).  NOT seen: in-place writes through
        ``p.data`` / ``p.detach()`` made after a training step -- call invalidate() after those."""
        return (self.param._version, self.raw_writes, sum(p._version for mem in self.members for p in mem))

    def invalidate(self):
        self.raw_writes += 1


def flatten_training_state(model, groups, device):
    """attach a FlatState to ``model`` (plans are rebuilt on the next call: parameter storage moved)."""
    model._flat = None
    model._plans = {}
    model._flat = FlatState(model, groups, device)
    model._plans = {}
    return model._flat


# ------------------------------------------------------------------------------------------------------
# execution: plan cache + autograd bridge
# ------------------------------------------------------------------------------------------------------
def _get_plan(model, N, H, W, training, device):
    L.lib()   # raises loudly if libpmf_amd.so is missing -- no fallback
    # BatchNorm modules switched to eval inside a training model (frozen statistics) select their own plan
    frozen = tuple(i for i, m in enumerate(model.modules())
                   if isinstance(m, nn.BatchNorm2d) and not m.training) if training else ()
    key = (N, H, W, bool(training), str(device), frozen)
    plan = model._plans.get(key)
    if plan is not None and plan.param_ptrs != [p.data_ptr() for p in plan.params]:
        plan = None    # parameters were re-allocated (e.g. .to()/.cuda()): rebuild
    if plan is None:
        with torch.cuda.device(device):
            plan = model._build(N, H, W, training, device)
        plan.generation = 0
        plan.bn_counters = [m.num_batches_tracked for m in plan.bn_modules if m.num_batches_tracked is not None]
        model._plans[key] = plan
    return plan


def _bind_io(plan, device):
    """plan-owned staging for everything that crosses the model boundary: inputs (NCHW, any strides / dtype -> one
    copy_), the probability maps the plan writes, and the upstream gradients of the backward pass.  Every pointer inside
    the op arrays is then FIXED for the life of the plan, so a captured hipGraph replays no matter where the caller's
    tensors live (a DataLoader + .cuda() loop hands over fresh addresses every iteration, trainer.py:289-303)."""
    plan.stage_in, plan.stage_out, plan.stage_g = {}, {}, {}
    for slot, idx in plan.in_slots.items():
        a = plan.fwd_ops[idx + plan.fwd_shift].u.sm
        n, c, hw = a.i[0], a.i[1], a.i[2]
        t = torch.zeros((n, c, hw), dtype=torch.float32, device=device)
        a.p[0] = t.data_ptr()
        a.l[0], a.l[1] = c * hw, hw
        plan.stage_in[slot] = t
    for slot, ent in plan.out_slots.items():
        o = torch.zeros(ent["shape"], dtype=torch.float32, device=device)
        plan.fwd_ops[ent["fwd_index"] + plan.fwd_shift].u.sm.p[1] = o.data_ptr()
        plan.stage_out[slot] = o
        if "bwd_index" in ent:
            g = torch.zeros(ent["shape"], dtype=torch.float32, device=device)
            b = plan.bwd_ops[ent["bwd_index"] + plan.bwd_shift].u.sm
            b.p[0], b.p[1] = o.data_ptr(), g.data_ptr()
            plan.stage_g[slot] = g


def _forward_impl(model, inputs):
    x0 = inputs[0]
    if not x0.is_cuda:
        raise RuntimeError("pmf_amd: the model runs only on an AMD GPU through libpmf_amd.so "
                           "(no CPU / PyTorch fallback); got a %s tensor" % x0.device)
    N, _, H, W = x0.shape
    plan = _get_plan(model, N, H, W, model.training, x0.device)
    if getattr(plan, "stage_in", None) is None:
        _bind_io(plan, x0.device)
    for nm, x in zip(("pcd", "rgb")[:len(inputs)], inputs):
        st = plan.stage_in[nm]
        st.view(x.shape).copy_(x)             # strided channel slices (trainer.py:296-297) / other dtypes: one copy
    if model.training:
        _fill_masks(plan, model)
    # weights already re-packed behind the optimiser update of the last step (engine.py _behind_events) and untouched since
    # (torch bumps the version counter of the flat buffer on every in-place write through a view): start behind the
    # forward-format pack launches
    begin = 0
    pv = getattr(plan, "packed_version", None)
    if pv is not None and plan.flat is not None and plan.flat.stamp() == pv:
        begin = plan.fwd_pack_skip
    plan.run(plan.fwd_ops, plan.n_fwd, "forward", begin, sig=())
    if model.training and plan.bn_counters:
        torch._foreach_add_(plan.bn_counters, 1)
    plan.generation += 1
    model._last_plan = plan            # (engine.py: the fused objective writes its gradient maps into this plan's staging buffers)
    # the caller owns what it gets: copies of the plan's probability maps (2 x 21 MB at 64 x 2048 bs 2: ~10 us)
    outs = [plan.stage_out[slot].clone() for slot in ("lidar", "camera") if slot in plan.out_slots]
    return plan, outs


class _PlanFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = forward plan, backward = backward plan.
    Parameters are explicit inputs so their gradients reach the leaf tensors (and DDP's reducer hooks)."""

    @staticmethod
    def forward(ctx, model, n_inputs, *args):
        inputs, params = args[:n_inputs], args[n_inputs:]
        plan, outs = _forward_impl(model, inputs)
        ctx.plan, ctx.generation, ctx.n_inputs, ctx.params = plan, plan.generation, n_inputs, params
        ctx.model = model
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("pmf_amd: backward() after a newer forward() on the same plan -- activations live "
                               "in the plan's arena; run forward/backward pairs in order")
        slots = [s for s in ("lidar", "camera") if s in plan.out_slots]
        for slot, g in zip(slots, gouts):
            if g is None:
                plan.stage_g[slot].zero_()
            elif not (g.data_ptr() == plan.stage_g[slot].data_ptr() and g.shape == plan.stage_g[slot].shape
                      and g.is_contiguous()):       # (the fused objective wrote straight into the staging buffer: nothing to copy)
                plan.stage_g[slot].copy_(g)
        sig = ()
        hook = getattr(ctx.model, "_bwd_segment_hook", None) if plan.flat is not None else None
        gated = getattr(ctx.model, "_bwd_gated_hook", None) if plan.flat is not None else None
        if gated is not None:
            # data parallel, default: the WHOLE backward plan as one range (same replay as the single-process step); the
            # hook then hangs the all-reduce of every gradient range behind the plan events that finalise it
            plan.run(plan.bwd_ops, plan.n_bwd, "backward", sig=sig)
            gated(plan)
        elif hook is None:
            plan.run(plan.bwd_ops, plan.n_bwd, "backward", sig=sig)
        else:       # data parallel (PMF_DP_MODE=segments): the backward plan in a few segments, finished ranges to the hook
            import os
            cuts = plan.segment_cuts(int(os.environ.get("PMF_DP_SEGMENTS", "4")))
            for k in range(len(cuts) - 1):
                plan.run(plan.bwd_ops, plan.n_bwd, "backward", cuts[k], cuts[k + 1], sig=sig)
                hook(plan, cuts[k + 1])
        if plan.flat is not None:
            # gradients were written in place into the FlatState buffer that every p.grad is a view of
            return (None, None) + (None,) * ctx.n_inputs + (None,) * len(ctx.params)
        # ONE device copy of the flat gradient buffer (the plan reuses it next iteration), then per-parameter views
        flat = plan.pgrad_buf.tensor((plan.pgrad_floats,)).clone()
        grads = []
        for p in ctx.params:
            ent = plan._pid.get(id(p))
            if ent is None:
                grads.append(None)
            else:
                grads.append(flat[ent[1]:ent[1] + p.numel()].view(p.shape))
        return (None, None) + (None,) * ctx.n_inputs + tuple(grads)


def _run_model(model, inputs):
    if not (torch.is_grad_enabled() and model.training):
        return tuple(_forward_impl(model, inputs)[1])
    params = [p for p in model.parameters() if p.requires_grad]
    return _PlanFunction.apply(model, len(inputs), *inputs, *params)
