"""Oracle: plain-PyTorch (CPU, fp32) restatement of the PMF network graph.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Every class cites the reference
lines it follows.  State-dict key names equal the reference's so the same
deterministic weights load into the reference modules, this oracle and the HIP
product model.

Differences from the reference that do not change arithmetic:
  * Dropout2d sites accept an injected [N, C] multiplier (``DropSite.mask``) so
    the HIP path and the oracle can be compared in train mode on the same masks
    (the reference draws them from torch's global RNG).
  * The torchvision ResNet bodies are restated here (torchvision is absent):
    BasicBlock / Bottleneck v1.5, layers [3,4,6,3] / [3,4,23,3] / [3,8,36,3].
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DropSite(nn.Module):
    """nn.Dropout2d(p) (salsanext.py:64,114-116,134; pmf_net.py:81) with optional injected mask."""

    def __init__(self, p=0.2):
        super().__init__()
        self.p = p
        self.mask = None  # [N, C] multiplier (0 or 1/(1-p)); None -> torch RNG

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        if self.mask is not None:
            return x * self.mask.to(x.dtype)[:, :, None, None]
        return F.dropout2d(x, self.p, True)


def _lrelu_bn(conv, bn, x):
    # SalsaNext ordering conv -> LeakyReLU(0.01) -> BN  (salsanext.py:27-33)
    return bn(F.leaky_relu(conv(x), 0.01))


class ResContextBlock(nn.Module):
    """salsanext.py:9-36."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv3 = nn.Conv2d(cout, cout, 3, dilation=2, padding=2)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        s = F.leaky_relu(self.conv1(x), 0.01)
        a1 = _lrelu_bn(self.conv2, self.bn1, s)
        a2 = _lrelu_bn(self.conv3, self.bn2, a1)
        return s + a2


class ResBlock(nn.Module):
    """salsanext.py:38-104.  Returns (pooled, skip) when pooling else the (dropped) tensor."""

    def __init__(self, cin, cout, p, pooling=True, drop_out=True):
        super().__init__()
        self.pooling, self.drop_out = pooling, drop_out
        self.conv1 = nn.Conv2d(cin, cout, 1)
        self.conv2 = nn.Conv2d(cin, cout, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv3 = nn.Conv2d(cout, cout, 3, dilation=2, padding=2)
        self.bn2 = nn.BatchNorm2d(cout)
        self.conv4 = nn.Conv2d(cout, cout, 2, dilation=2, padding=1)
        self.bn3 = nn.BatchNorm2d(cout)
        self.conv5 = nn.Conv2d(3 * cout, cout, 1)
        self.bn4 = nn.BatchNorm2d(cout)
        self.dropout = DropSite(p)

    def forward(self, x):
        s = F.leaky_relu(self.conv1(x), 0.01)
        r1 = _lrelu_bn(self.conv2, self.bn1, x)
        r2 = _lrelu_bn(self.conv3, self.bn2, r1)
        r3 = _lrelu_bn(self.conv4, self.bn3, r2)
        a = s + _lrelu_bn(self.conv5, self.bn4, torch.cat((r1, r2, r3), 1))
        b = self.dropout(a) if self.drop_out else a
        if self.pooling:
            return F.avg_pool2d(b, 3, 2, 1), a  # count_include_pad=True
        return b


class UpBlock(nn.Module):
    """salsanext.py:107-164."""

    def __init__(self, cin, cout, p, drop_out=True):
        super().__init__()
        self.drop_out = drop_out
        self.dropout1, self.dropout2, self.dropout3 = DropSite(p), DropSite(p), DropSite(p)
        self.conv1 = nn.Conv2d(cin // 4 + 2 * cout, cout, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, dilation=2, padding=2)
        self.bn2 = nn.BatchNorm2d(cout)
        self.conv3 = nn.Conv2d(cout, cout, 2, dilation=2, padding=1)
        self.bn3 = nn.BatchNorm2d(cout)
        self.conv4 = nn.Conv2d(3 * cout, cout, 1)
        self.bn4 = nn.BatchNorm2d(cout)

    def forward(self, x, skip):
        a = F.pixel_shuffle(x, 2)
        if self.drop_out:
            a = self.dropout1(a)
        b = torch.cat((a, skip), 1)
        if self.drop_out:
            b = self.dropout2(b)
        e1 = _lrelu_bn(self.conv1, self.bn1, b)
        e2 = _lrelu_bn(self.conv2, self.bn2, e1)
        e3 = _lrelu_bn(self.conv3, self.bn3, e2)
        e = _lrelu_bn(self.conv4, self.bn4, torch.cat((e1, e2, e3), 1))
        return self.dropout3(e) if self.drop_out else e


class SalsaNext(nn.Module):
    """salsanext.py:166-208."""

    def __init__(self, in_channels=8, nclasses=20, base_channels=32, softmax=True):
        super().__init__()
        c, p = base_channels, 0.2
        self.base_channels, self.softmax = c, softmax
        self.downCntx = ResContextBlock(in_channels, c)
        self.downCntx2 = ResContextBlock(c, c)
        self.downCntx3 = ResContextBlock(c, c)
        self.resBlock1 = ResBlock(c, 2 * c, p, True, False)
        self.resBlock2 = ResBlock(2 * c, 4 * c, p, True)
        self.resBlock3 = ResBlock(4 * c, 8 * c, p, True)
        self.resBlock4 = ResBlock(8 * c, 8 * c, p, True)
        self.resBlock5 = ResBlock(8 * c, 8 * c, p, False)
        self.upBlock1 = UpBlock(8 * c, 4 * c, p)
        self.upBlock2 = UpBlock(4 * c, 4 * c, p)
        self.upBlock3 = UpBlock(4 * c, 2 * c, p)
        self.upBlock4 = UpBlock(2 * c, c, p, False)
        self.logits = nn.Conv2d(c, nclasses, 1)

    def _fuse(self, i, x, feats):
        return x

    def _bottleneck(self, x):
        return x

    def forward(self, x, img_feature=()):
        d = self.downCntx3(self.downCntx2(self.downCntx(x)))
        d0c, d0b = self.resBlock1(d)
        d0c = self._fuse(1, d0c, img_feature)
        d1c, d1b = self.resBlock2(d0c)
        d1c = self._fuse(2, d1c, img_feature)
        d2c, d2b = self.resBlock3(d1c)
        d2c = self._fuse(3, d2c, img_feature)
        d3c, d3b = self.resBlock4(d2c)
        d3c = self._fuse(4, d3c, img_feature)
        d5c = self._bottleneck(self.resBlock5(d3c))
        u = self.upBlock1(d5c, d3b)
        u = self.upBlock2(u, d2b)
        u = self.upBlock3(u, d1b)
        u = self.upBlock4(u, d0b)
        self.last_logits = self.logits(u)  # pre-softmax hook point (SURVEY 7 "tolerance definition")
        return F.softmax(self.last_logits, 1) if self.softmax else self.last_logits


class ResidualBasedFusionBlock(nn.Module):
    """pmf_net.py:10-36."""

    def __init__(self, pcd_channels, img_channels):
        super().__init__()
        p = pcd_channels
        self.fuse_conv = nn.Sequential(
            nn.Conv2d(p + img_channels, p, 3, padding=1), nn.LeakyReLU(), nn.BatchNorm2d(p))
        self.attention = nn.Sequential(
            nn.Conv2d(p, p, 3, padding=1), nn.BatchNorm2d(p), nn.ReLU(),
            nn.Conv2d(p, p, 3, padding=1), nn.BatchNorm2d(p), nn.Sigmoid())

    def forward(self, pcd, img):
        f = self.fuse_conv(torch.cat((pcd, img), 1))
        return f * self.attention(f) + pcd


class ASPP(nn.Module):
    """pmf_net.py:103-138 (no BN, no activation)."""

    def __init__(self, in_channel=512, depth=256):
        super().__init__()
        self.conv = nn.Conv2d(in_channel, depth, 1)
        self.atrous_block1 = nn.Conv2d(in_channel, depth, 1)
        self.atrous_block6 = nn.Conv2d(in_channel, depth, 3, padding=6, dilation=6)
        self.atrous_block12 = nn.Conv2d(in_channel, depth, 3, padding=12, dilation=12)
        self.atrous_block18 = nn.Conv2d(in_channel, depth, 3, padding=18, dilation=18)
        self.conv_1x1_output = nn.Conv2d(depth * 5, depth, 1)

    def forward(self, x):
        g = self.conv(x.mean((2, 3), keepdim=True)).expand(-1, -1, x.shape[2], x.shape[3])
        return self.conv_1x1_output(torch.cat(
            (g, self.atrous_block1(x), self.atrous_block6(x),
             self.atrous_block12(x), self.atrous_block18(x)), 1))


class SalsaNextFusion(SalsaNext):
    """pmf_net.py:141-180."""

    def __init__(self, in_channels=8, nclasses=20, base_channels=32, img_feature_channels=()):
        super().__init__(in_channels, nclasses, base_channels, True)
        c = base_channels
        for i, pc in enumerate((2 * c, 4 * c, 8 * c, 8 * c)):
            setattr(self, "fusionblock_%d" % (i + 1),
                    ResidualBasedFusionBlock(pc, img_feature_channels[i]))
        self.aspp = ASPP(8 * c, 8 * c)

    def _fuse(self, i, x, feats):
        return getattr(self, "fusionblock_%d" % i)(x, feats[i - 1])

    def _bottleneck(self, x):
        return self.aspp(x)


# ---- torchvision.models.resnet restatement (third-party; parity unpinned) -------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + idt)


RESNET_CFG = {"resnet34": (BasicBlock, (3, 4, 6, 3)), "resnet50": (Bottleneck, (3, 4, 6, 3)),
              "resnet101": (Bottleneck, (3, 4, 23, 3)), "resnet152": (Bottleneck, (3, 8, 36, 3))}


def _make_layer(block, cin, planes, n, stride):
    ds = None
    if stride != 1 or cin != planes * block.expansion:
        ds = nn.Sequential(nn.Conv2d(cin, planes * block.expansion, 1, stride, bias=False),
                           nn.BatchNorm2d(planes * block.expansion))
    layers = [block(cin, planes, stride, ds)]
    layers += [block(planes * block.expansion, planes) for _ in range(n - 1)]
    return nn.Sequential(*layers)


class ResNet(nn.Module):
    """pmf_net.py:41-100: torchvision body with a stride-1 7x7 stem and Dropout2d on layer3/4."""

    def __init__(self, in_channels=3, backbone="resnet50", dropout_rate=0.2, pretrained=False):
        super().__init__()
        if backbone not in RESNET_CFG:
            raise NotImplementedError("invalid backbone: {}".format(backbone))
        block, counts = RESNET_CFG[backbone]
        self.expansion = block.expansion
        e = self.expansion
        self.feature_channels = [64 * e, 128 * e, 256 * e, 512 * e]
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 1, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _make_layer(block, 64, 64, counts[0], 1)
        self.layer2 = _make_layer(block, 64 * e, 128, counts[1], 2)
        self.layer3 = _make_layer(block, 128 * e, 256, counts[2], 2)
        self.layer4 = _make_layer(block, 256 * e, 512, counts[3], 2)
        self.dropout = DropSite(dropout_rate)   # ONE module applied twice (pmf_net.py:97-98)
        self.dropout_b = None                   # second injected mask for the layer4 application
        for m in self.modules():                # torchvision init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        h, w = x.shape[2], x.shape[3]
        if h % 16 != 0 or w % 16 != 0:
            assert False, "invalid input size: {}".format(x.shape)
        c1 = F.relu(self.bn1(self.conv1(x)))
        f0 = self.layer1(F.max_pool2d(c1, 3, 2, 1))
        f1 = self.layer2(f0)
        f2 = self.dropout(self.layer3(f1))
        if self.dropout_b is not None:
            keep, self.dropout.mask = self.dropout.mask, self.dropout_b
            f3 = self.dropout(self.layer4(f2))
            self.dropout.mask = keep
        else:
            f3 = self.dropout(self.layer4(f2))
        return [f0, f1, f2, f3]


class RGBDecoder(nn.Module):
    """pmf_net.py:183-222."""

    def __init__(self, in_channels=(), nclasses=4, base_channels=64):
        super().__init__()
        b = base_channels

        def up(cin, k):
            return nn.Sequential(nn.Conv2d(cin, b, k, padding=k // 2), nn.LeakyReLU(), nn.BatchNorm2d(b),
                                 nn.Upsample(scale_factor=2, mode="bilinear"))
        self.up_4a = up(in_channels[3], 3)
        self.up_3a = up(in_channels[2] + b, 3)
        self.up_2a = up(in_channels[1] + b, 3)
        self.up_1a = up(in_channels[0] + b, 1)
        self.conv = nn.Conv2d(b, nclasses, 3, padding=1)

    def forward(self, f):
        u = self.up_4a(f[3])
        u = self.up_3a(torch.cat((u, f[2]), 1))
        u = self.up_2a(torch.cat((u, f[1]), 1))
        u = self.up_1a(torch.cat((u, f[0]), 1))
        self.last_logits = self.conv(u)
        return F.softmax(self.last_logits, 1)


class PMFNet(nn.Module):
    """pmf_net.py:224-249."""

    def __init__(self, pcd_channels=5, img_channels=3, nclasses=20, base_channels=32,
                 imagenet_pretrained=False, image_backbone="resnet34"):
        super().__init__()
        self.camera_stream_encoder = ResNet(img_channels, image_backbone, pretrained=imagenet_pretrained)
        self.camera_stream_decoder = RGBDecoder(
            self.camera_stream_encoder.feature_channels, nclasses,
            self.camera_stream_encoder.expansion * 16)
        self.lidar_stream = SalsaNextFusion(pcd_channels, nclasses, base_channels,
                                            self.camera_stream_encoder.feature_channels)

    def forward(self, pcd_feature, img_feature):
        feats = self.camera_stream_encoder(img_feature)
        lidar_pred = self.lidar_stream(pcd_feature, feats)
        camera_pred = self.camera_stream_decoder(feats)
        return lidar_pred, camera_pred


# ---- dropout-mask plumbing shared with the HIP product (same site order) -------------------
def dropout_sites(model):
    """(name, module, channels) for every active Dropout2d application, in forward order.

    The encoder's single Dropout2d module is applied to layer3 and layer4 outputs
    (pmf_net.py:97-98), hence two entries ("enc.f2", "enc.f3")."""
    enc, ls = model.camera_stream_encoder, model.lidar_stream
    c = ls.base_channels
    sites = [("enc.f2", enc, enc.feature_channels[2]), ("enc.f3", enc, enc.feature_channels[3])]
    for i, ch in ((2, 4 * c), (3, 8 * c), (4, 8 * c), (5, 8 * c)):
        sites.append(("resBlock%d" % i, getattr(ls, "resBlock%d" % i).dropout, ch))
    for i, (cin, cout) in ((1, (8 * c, 4 * c)), (2, (4 * c, 4 * c)), (3, (4 * c, 2 * c))):
        ub = getattr(ls, "upBlock%d" % i)
        sites.append(("upBlock%d.d1" % i, ub.dropout1, cin // 4))
        sites.append(("upBlock%d.d2" % i, ub.dropout2, cin // 4 + 2 * cout))
        sites.append(("upBlock%d.d3" % i, ub.dropout3, cout))
    return sites


def set_dropout_masks(model, masks):
    """masks: dict name -> [N, C] tensor (0 or 1/(1-p)); None clears."""
    enc = model.camera_stream_encoder
    for name, mod, _ in dropout_sites(model):
        m = None if masks is None else masks[name]
        if name == "enc.f2":
            enc.dropout.mask = m
        elif name == "enc.f3":
            enc.dropout_b = m
        else:
            mod.mask = m
