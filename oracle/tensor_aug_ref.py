"""Oracle: torch-CPU restatement of the perspective loader's training-time tensor transforms (TEST INFRASTRUCTURE).

pc_processor/dataset/perspective_view_loader.py:63-69,138-141 composes torchvision transforms (third party, v0.14.1 per
README_en.md:73, NOT installed here -> **parity unpinned**): RandomHorizontalFlip(0.5), RandomRotation(15),
RandomCrop(size), then Pad.  Restated from torchvision's published tensor path:
  * draws (torch global RNG, in this order): flip if torch.rand(1) < p; angle = torch.empty(1).uniform_(-deg, deg);
    crop offsets i = torch.randint(0, h-th+1, (1,)), j = torch.randint(0, w-tw+1, (1,)) (no draw when sizes are equal);
  * rotate(img, angle): inverse affine matrix [cos r, sin r, 0, -sin r, cos r, 0] with r = radians(-angle); affine
    grid over pixel centres (x in linspace(-w/2+0.5, w/2-0.5, w)), divided by (w/2, h/2), sampled with
    torch.nn.functional.grid_sample(mode="nearest", padding_mode="zeros", align_corners=False) -- torch's own kernel, so
    the nearest-pixel rounding here IS PyTorch's; zero fill outside."""
import math

import torch
import torch.nn.functional as F


def draw_params(h, w, crop_h, crop_w, p=0.5, degrees=15.0):
    flip = bool(torch.rand(1) < p)
    angle = float(torch.empty(1).uniform_(-float(degrees), float(degrees)).item())
    if h < crop_h or w < crop_w:
        raise ValueError("Required crop size {} is larger than input image size {}".format((crop_h, crop_w), (h, w)))
    if (h, w) == (crop_h, crop_w):
        top, left = 0, 0
    else:
        top = int(torch.randint(0, h - crop_h + 1, size=(1,)).item())
        left = int(torch.randint(0, w - crop_w + 1, size=(1,)).item())
    return flip, angle, top, left


def inverse_rotation_matrix(angle):
    r = math.radians(-angle)
    return [math.cos(r), math.sin(r), 0.0, -math.sin(r), math.cos(r), 0.0]


def rotate_nearest(img, angle):
    c, h, w = img.shape
    theta = torch.tensor(inverse_rotation_matrix(angle), dtype=img.dtype).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    return F.grid_sample(img[None], grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]


def flip_rotate_crop(img, flip, angle, top, left, crop_h, crop_w, h_pad=0, w_pad=0):
    x = img.flip(-1) if flip else img
    x = rotate_nearest(x, angle)
    x = x[:, top:top + crop_h, left:left + crop_w]
    return F.pad(x, (w_pad, w_pad, h_pad, h_pad))
