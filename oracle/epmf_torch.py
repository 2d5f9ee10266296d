"""TEST INFRASTRUCTURE -- CPU restatement of the reference's EPMFNet (pc_processor/models/epmf_net.py) with the
reference's parameter names.  Imported only by tests/, oracle/make_golden.py and __graft_entry__.smoke(); the product
(pmf_amd/models/epmf_net.py) never touches it.  Pinned against tests/golden/g8_epmf.npz, which oracle/make_golden.py
produces by running the reference's own epmf_net.py (camera backbone: torchvision stand-in, see pmf_torch.py).

Restated pieces cite the reference lines they follow; blocks shared with PMF come from oracle/pmf_torch.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pmf_torch import ASPP, ResidualBasedFusionBlock, ResNet, SalsaNext


class SparseVariantConv(nn.Module):
    """epmf_net.py:10-50.  The ``mask_conv`` normaliser (:33-40) is computed and discarded by the reference; omitted."""

    def __init__(self, cin, cout, kernel_size, padding=0, stride=1, groups=1, dilation=1, bias=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=kernel_size, padding=padding, stride=stride, groups=groups,
                              dilation=dilation)
        self.pool = nn.MaxPool2d(kernel_size, stride=stride, padding=0, dilation=dilation)
        self.bias = nn.Parameter(torch.zeros(cout).float()) if bias else None

    def forward(self, x, mask):
        x = x * mask                                                          # :31
        with torch.no_grad():
            ph, pw = self.conv.padding
            mask = self.pool(F.pad(mask, (pw, pw, ph, ph)))                   # :41-43 dilated mask
        x = self.conv(x)
        if self.bias is not None:
            x = x + self.bias.view(1, -1, 1, 1)                               # :46-47
        return x * mask, mask                                                 # :48


class ResContextBlock(nn.Module):
    """epmf_net.py:52-80."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = SparseVariantConv(cin, cout, 3, padding=1, stride=stride)
        self.act1 = nn.LeakyReLU()
        self.conv2 = SparseVariantConv(cout, cout, (3, 3), padding=(1, 1))
        self.act2 = nn.LeakyReLU()
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv3 = SparseVariantConv(cout, cout, (3, 3), padding=(2, 2), dilation=2)
        self.act3 = nn.LeakyReLU()
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        mask = x.abs().sum(1).ne(0).to(x.dtype).unsqueeze(1)                  # :67
        shortcut, mask = self.conv1(x, mask)
        shortcut = self.act1(shortcut)
        a, mask = self.conv2(shortcut, mask)
        a1 = self.bn1(self.act2(a))
        a, mask = self.conv3(a1, mask)
        a2 = self.bn2(self.act3(a))
        return (shortcut + a2) * mask                                         # :78-79


class SalsaNextFusion(SalsaNext):
    """epmf_net.py:82-131."""

    def __init__(self, in_channels=8, nclasses=20, base_channels=32, img_feature_channels=()):
        super().__init__(in_channels, nclasses, base_channels, True)
        c = base_channels
        self.downCntx = ResContextBlock(in_channels, c)
        self.downCntx2 = ResContextBlock(c, c)
        self.downCntx3 = ResContextBlock(c, c, stride=2)
        self.fusionblock_1 = ResidualBasedFusionBlock(c * 1, img_feature_channels[0])
        self.fusionblock_2 = ResidualBasedFusionBlock(c * 2, img_feature_channels[1])
        self.fusionblock_3 = ResidualBasedFusionBlock(c * 4, img_feature_channels[2])
        self.fusionblock_4 = ResidualBasedFusionBlock(c * 8, img_feature_channels[3])
        self.aspp = ASPP(c * 8, c * 8)
        self.extraUpSample = nn.Sequential(nn.Conv2d(c, 4 * c, 3, padding=1), nn.LeakyReLU(), nn.BatchNorm2d(4 * c),
                                           nn.PixelShuffle(2))

    def forward(self, x, img_feature=()):
        d = self.downCntx3(self.downCntx2(self.downCntx(x)))
        d = self.fusionblock_1(d, img_feature[0])
        d0c, d0b = self.resBlock1(d)
        d0c = self.fusionblock_2(d0c, img_feature[1])
        d1c, d1b = self.resBlock2(d0c)
        d1c = self.fusionblock_3(d1c, img_feature[2])
        d2c, d2b = self.resBlock3(d1c)
        d2c = self.fusionblock_4(d2c, img_feature[3])
        d3c, d3b = self.resBlock4(d2c)
        d5c = self.aspp(self.resBlock5(d3c))
        u = self.upBlock1(d5c, d3b)
        u = self.upBlock2(u, d2b)
        u = self.upBlock3(u, d1b)
        u = self.upBlock4(u, d0b)
        u = self.extraUpSample(u)
        self.last_logits = self.logits(u)
        return F.softmax(self.last_logits, 1), d5c


class RGBDecoder(nn.Module):
    """epmf_net.py:134-183."""

    def __init__(self, in_channels=(), nclasses=4, base_channels=64, lidar_base_channels=32):
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
        self.conv = nn.Conv2d(b, nclasses, 3, padding=1)

    def forward(self, f, lidar_feature):
        u = self.up_4a(torch.cat((self.extraUpSample(lidar_feature), self.aspp(f[3])), 1))
        u = self.up_3a(torch.cat((u, f[2]), 1))
        u = self.up_2a(torch.cat((u, f[1]), 1))
        u = self.up_1a(torch.cat((u, f[0]), 1))
        self.last_logits = self.conv(u)
        return F.softmax(self.last_logits, 1)


class EPMFNet(nn.Module):
    """epmf_net.py:185-215."""

    def __init__(self, pcd_channels=5, img_channels=3, nclasses=20, base_channels=32, imagenet_pretrained=False,
                 image_backbone="resnet34"):
        super().__init__()
        self.camera_stream_encoder = ResNet(img_channels, image_backbone, pretrained=imagenet_pretrained)
        self.camera_stream_decoder = RGBDecoder(self.camera_stream_encoder.feature_channels, nclasses,
                                                self.camera_stream_encoder.expansion * 16, base_channels)
        self.lidar_stream = SalsaNextFusion(pcd_channels, nclasses, base_channels,
                                            self.camera_stream_encoder.feature_channels)

    def forward(self, pcd_feature, img_feature):
        feats = self.camera_stream_encoder(img_feature)
        lidar_pred, lidar_feature = self.lidar_stream(pcd_feature, feats)
        return lidar_pred, self.camera_stream_decoder(feats, lidar_feature)


class MultiTaskLoss(nn.Module):
    """pc_processor/loss/multi_task_loss.py:5-19: sum_i L_i / (2 sigma_i^2) + log(sigma_i^2 + 1)."""

    def __init__(self, n_losses, sigma=None):
        super().__init__()
        self.sigma = nn.Parameter(torch.Tensor(sigma) if sigma is not None else torch.ones(n_losses) / n_losses)

    def forward(self, losses):
        total = 0
        for i, l in enumerate(losses):
            total = total + l / (2.0 * self.sigma[i].pow(2)) + (self.sigma[i].pow(2) + 1.0).log()
        return total
