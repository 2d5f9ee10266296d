"""EPMFNet on MI355X (pc_processor/models/epmf_net.py): PMF's dual-branch network with
  * SparseVariantConv context blocks (input * validity mask, (conv + two biases) * dilated mask; the ``mask_conv``
    normaliser of the reference is dead code, epmf_net.py:33-40, and is not computed),
  * a stride-2 third context block (the whole LiDAR trunk runs at half resolution), fusion BEFORE each resBlock,
  * ``extraUpSample`` (3x3 conv -> LeakyReLU -> BN -> PixelShuffle) back to full resolution,
  * the LiDAR bottleneck feature fed into the camera decoder.
Same contract as PMFNet: the module tree only holds parameters (identical state-dict keys), forward is one static
HIP plan, no CPU fallback.  New kernels: per-pixel masks (elementwise.hip pmask_*), conv epilogue ``ep_pmask``."""
import torch
import torch.nn as nn

from .. import _lib as L
from ..plan import Plan, V
from .pmf_net import (_Holder, _alloc_masks, _run_model, _salsa_sites, ASPP, ResidualBasedFusionBlock, ResNet,
                      SalsaNext)

__all__ = ["EPMFNet", "SparseVariantConv", "EPMFResContextBlock", "EPMFSalsaNextFusion", "EPMFRGBDecoder"]


class SparseVariantConv(_Holder):
    """epmf_net.py:10-50: holds ``conv`` (nn.Conv2d WITH its own bias), ``pool`` and the extra ``bias`` parameter."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, stride=1, groups=1, dilation=1, bias=True):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped SparseVariantConv is not used by EPMF")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=padding, stride=stride,
                              groups=groups, dilation=dilation)
        self.pool = nn.MaxPool2d(kernel_size, stride=stride, padding=0, dilation=dilation)
        self.bias = nn.Parameter(torch.zeros(out_channels).float()) if bias else None
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out", nonlinearity="leaky_relu")

    def emit(self, P, x, pm, act, bn, name):
        """x: V, pm: validity mask of x.  Returns (V of act((conv(x*pm)+b)*pm'), [BN view], pm')."""
        xm = V(P.pmask_mul(x, pm, name + ".in"))
        pm2 = P.pmask_pool(pm, self.conv)
        y = P.conv([xm], self.conv, act, bn, name=name, pmask=pm2, extra_bias=self.bias)
        return y, pm2


class EPMFResContextBlock(_Holder):
    """epmf_net.py:52-80."""

    def __init__(self, in_filters, out_filters, stride=1):
        super().__init__()
        self.conv1 = SparseVariantConv(in_filters, out_filters, 3, padding=1, stride=stride)
        self.act1 = nn.LeakyReLU()
        self.conv2 = SparseVariantConv(out_filters, out_filters, (3, 3), padding=(1, 1))
        self.act2 = nn.LeakyReLU()
        self.bn1 = nn.BatchNorm2d(out_filters)
        self.conv3 = SparseVariantConv(out_filters, out_filters, (3, 3), padding=(2, 2), dilation=2)
        self.act3 = nn.LeakyReLU()
        self.bn2 = nn.BatchNorm2d(out_filters)

    def emit(self, P, x, name):
        pm = P.pmask_from(x)
        s, pm = self.conv1.emit(P, x, pm, L.ACT_LRELU, None, name + ".s")
        a1, pm = self.conv2.emit(P, s, pm, L.ACT_LRELU, self.bn1, name + ".a1")
        a2, pm = self.conv3.emit(P, a1, pm, L.ACT_LRELU, self.bn2, name + ".a2")
        out = P.add_act(s, a2, L.ACT_NONE, name=name + ".sum")
        return V(P.pmask_mul(V(out), pm, name + ".out"))


class EPMFSalsaNextFusion(SalsaNext):
    """epmf_net.py:82-131."""

    def __init__(self, in_channels=8, nclasses=20, base_channels=32, img_feature_channels=[]):
        super().__init__(in_channels=in_channels, base_channels=base_channels, nclasses=nclasses, softmax=True)
        c = self.base_channels
        self.downCntx = EPMFResContextBlock(in_channels, c)
        self.downCntx2 = EPMFResContextBlock(c, c)
        self.downCntx3 = EPMFResContextBlock(c, c, stride=2)
        self.fusionblock_1 = ResidualBasedFusionBlock(c * 1, img_feature_channels[0])
        self.fusionblock_2 = ResidualBasedFusionBlock(c * 2, img_feature_channels[1])
        self.fusionblock_3 = ResidualBasedFusionBlock(c * 4, img_feature_channels[2])
        self.fusionblock_4 = ResidualBasedFusionBlock(c * 8, img_feature_channels[3])
        self.aspp = ASPP(c * 8, c * 8)
        self.extraUpSample = nn.Sequential(nn.Conv2d(c, 4 * c, 3, padding=1), nn.LeakyReLU(), nn.BatchNorm2d(4 * c),
                                           nn.PixelShuffle(2))

    def emit_trunk(self, P, x, feats, M):
        m = (lambda k: M[k]) if M is not None else (lambda k: None)
        d = self.downCntx.emit(P, x, "downCntx")
        d = self.downCntx2.emit(P, d, "downCntx2")
        d = self.downCntx3.emit(P, d, "downCntx3")
        d = self.fusionblock_1.emit(P, d.t, feats[0], "fusion1")
        d0c, d0b = self.resBlock1.emit(P, V(d), "resBlock1")
        d0c = self.fusionblock_2.emit(P, d0c, feats[1], "fusion2")
        d1c, d1b = self.resBlock2.emit(P, V(d0c), "resBlock2", m("resBlock2"))
        d1c = self.fusionblock_3.emit(P, d1c, feats[2], "fusion3")
        d2c, d2b = self.resBlock3.emit(P, V(d1c), "resBlock3", m("resBlock3"))
        d2c = self.fusionblock_4.emit(P, d2c, feats[3], "fusion4")
        d3c, d3b = self.resBlock4.emit(P, V(d2c), "resBlock4", m("resBlock4"))
        d5c = self.aspp.emit(P, self.resBlock5.emit(P, V(d3c), "resBlock5", m("resBlock5")), "aspp")
        um = (lambda i: dict(comb=M["upBlock%d.comb" % i], d2=M["upBlock%d.d2" % i], d3=M["upBlock%d.d3" % i])) \
            if M is not None else (lambda i: None)
        u = self.upBlock1.emit(P, d5c, d3b, "upBlock1", um(1))
        u = self.upBlock2.emit(P, u, d2b, "upBlock2", um(2))
        u = self.upBlock3.emit(P, u, d1b, "upBlock3", um(3))
        u = self.upBlock4.emit(P, u, d0b, "upBlock4", None)
        e = P.conv([u], self.extraUpSample[0], L.ACT_LRELU, self.extraUpSample[2], name="extraUp")
        up = P.pixel_shuffle(e, None, e.t.C // 4, name="extraUp.ps")
        lg = P.conv([V(up)], self.logits, name="logits")
        P.softmax_out(lg.t, "lidar", "lidar")
        return d5c

    def forward(self, *a, **k):
        raise RuntimeError("EPMFSalsaNextFusion runs inside EPMFNet's plan; call EPMFNet")


class EPMFRGBDecoder(_Holder):
    """epmf_net.py:134-183."""

    def __init__(self, in_channels=[], nclasses=4, base_channels=64, lidar_base_channels=32):
        super().__init__()
        b, lb = base_channels, lidar_base_channels
        self.aspp = ASPP(in_channels[3], in_channels[3])
        self.extraUpSample = nn.Sequential(nn.Conv2d(lb * 8, lb * 8, 3, padding=1), nn.LeakyReLU(),
                                           nn.BatchNorm2d(lb * 8), nn.PixelShuffle(2))

        def up(cin, k):
            return nn.Sequential(nn.Conv2d(cin, b, k, padding=k // 2), nn.LeakyReLU(), nn.BatchNorm2d(b),
                                 nn.Upsample(scale_factor=2, mode="bilinear"))
        self.up_4a = up(in_channels[3] + lb * 2, 3)
        self.up_3a = up(in_channels[2] + b, 3)
        self.up_2a = up(in_channels[1] + b, 3)
        self.up_1a = up(in_channels[0] + b, 1)
        self.conv = nn.Conv2d(b, nclasses, kernel_size=3, padding=1)

    def emit(self, P, feats, lidar_feature):
        def up(seq, srcs, name):
            v = P.conv(srcs, seq[0], L.ACT_LRELU, seq[2], name=name)
            return V(P.bilinear(v, name=name + ".up"))
        e = P.conv([lidar_feature], self.extraUpSample[0], L.ACT_LRELU, self.extraUpSample[2], name="dec.extraUp")
        lf = P.pixel_shuffle(e, None, e.t.C // 4, name="dec.extraUp.ps")
        a = self.aspp.emit(P, feats[3], "dec.aspp")
        u = up(self.up_4a, [V(lf), a], "dec.up4")
        u = up(self.up_3a, [u, feats[2]], "dec.up3")
        u = up(self.up_2a, [u, feats[1]], "dec.up2")
        u = up(self.up_1a, [u, feats[0]], "dec.up1")
        lg = P.conv([u], self.conv, name="dec.logits")
        P.softmax_out(lg.t, "camera", "camera")
        return lg


class EPMFNet(nn.Module):
    """epmf_net.py:185-215 -- same call surface as PMFNet; H and W must be multiples of 32 (half-resolution trunk)."""

    def __init__(self, pcd_channels=5, img_channels=3, nclasses=20, base_channels=32, imagenet_pretrained=True,
                 image_backbone="resnet34"):
        super().__init__()
        if "resnet" not in image_backbone:
            raise NotImplementedError(image_backbone)
        self.camera_stream_encoder = ResNet(in_channels=img_channels, pretrained=imagenet_pretrained,
                                            backbone=image_backbone)
        self.camera_stream_decoder = EPMFRGBDecoder(self.camera_stream_encoder.feature_channels, nclasses=nclasses,
                                                    base_channels=self.camera_stream_encoder.expansion * 16,
                                                    lidar_base_channels=base_channels)
        self.lidar_stream = EPMFSalsaNextFusion(in_channels=pcd_channels, nclasses=nclasses,
                                                base_channels=base_channels,
                                                img_feature_channels=self.camera_stream_encoder.feature_channels)
        self.pcd_channels, self.img_channels, self.nclasses = pcd_channels, img_channels, nclasses
        self._plans = {}

    def forward(self, pcd_feature, img_feature):
        h, w = img_feature.shape[2], img_feature.shape[3]
        if h % 32 != 0 or w % 32 != 0:
            assert False, "invalid input size: {}".format(img_feature.shape)
        if pcd_feature.shape[1] != self.pcd_channels or img_feature.shape[1] != self.img_channels or \
                pcd_feature.shape[0] != img_feature.shape[0] or pcd_feature.shape[2:] != img_feature.shape[2:]:
            raise ValueError("EPMFNet: expected [N,%d,H,W] and [N,%d,H,W] inputs, got %s and %s" % (
                self.pcd_channels, self.img_channels, tuple(pcd_feature.shape), tuple(img_feature.shape)))
        return _run_model(self, (pcd_feature, img_feature))

    def _build(self, N, H, W, training, device, dry=False):
        P = Plan(device, training, getattr(self, "_flat", None), dry)
        M = _alloc_masks(P, self, N, device) if training else None
        pcd = V(P.input_nchw("pcd", N, self.pcd_channels, H, W, "pcd"))
        rgb = V(P.input_nchw("rgb", N, self.img_channels, H, W, "rgb"))
        feats = self.camera_stream_encoder.emit(P, rgb, M)
        lidar_feature = self.lidar_stream.emit_trunk(P, pcd, feats, M)
        self.camera_stream_decoder.emit(P, feats, lidar_feature)
        return P.finalise()

    def _mask_sites(self):
        enc = self.camera_stream_encoder
        return [("enc.f2", enc.feature_channels[2]), ("enc.f3", enc.feature_channels[3])] + \
            _salsa_sites(self.lidar_stream)

    def _apply(self, fn, *a, **k):
        self._plans = {}
        self._flat = None
        return super()._apply(fn, *a, **k)

    def set_dropout_masks(self, masks):
        self._forced_masks = masks
