"""Oracle: numpy restatement of the image jitter of the perspective loaders (TEST INFRASTRUCTURE).

pc_processor/dataset/perspective_view_loader.py:46-49,84-85 and perspective_view_loader_v2.py:19-23,46-47 apply
``torchvision.transforms.ColorJitter(*img_jitter)`` to the PIL image before anything else.  torchvision (0.14.1,
README_en.md:73) is a third-party dependency that is not installed; its PIL path is a thin layer over Pillow, which IS
installed, so this restatement is pinned in two steps (tests/test_oracle_golden.py):
  * the four pixel operations against Pillow itself (ImageEnhance.Brightness / Contrast / Color = Image.blend with a
    degenerate image; the hue shift = convert("HSV"), uint8 add on H, convert back), exhaustively over all 2^24 colours
    for the colour-space maps and all 2^16 operand pairs for the blend;
  * the parameter draw order against torchvision's published ``ColorJitter.get_params`` (restated below; unpinned).

torchvision ColorJitter(brightness, contrast, saturation, hue), PIL input:
    ranges: brightness/contrast/saturation v -> [max(0, 1 - v), 1 + v], hue v -> [-v, v]; a range whose two ends equal the
    centre (v == 0) is None and draws nothing.
    get_params (torch global RNG, this order): fn_idx = torch.randperm(4); then for brightness, contrast, saturation,
    hue in turn: float(torch.empty(1).uniform_(lo, hi)) unless the range is None.
    forward: for fn_id in fn_idx: 0 brightness, 1 contrast, 2 saturation, 3 hue -- each on the uint8 result of the
    previous one.
"""
import numpy as np
import torch


def jitter_ranges(brightness=0, contrast=0, saturation=0, hue=0):
    def rng(v, center, clip_first_on_zero):
        if isinstance(v, (tuple, list)):
            lo, hi = float(v[0]), float(v[1])
        else:
            lo, hi = center - float(v), center + float(v)
            if clip_first_on_zero:
                lo = max(lo, 0.0)
        return None if lo == hi == center else (lo, hi)
    return (rng(brightness, 1.0, True), rng(contrast, 1.0, True), rng(saturation, 1.0, True), rng(hue, 0.0, False))


def draw_params(ranges):
    """(order int[4], factors [b, c, s, h] with None for unused) drawn from torch's global RNG in torchvision's order."""
    order = torch.randperm(4).tolist()
    fac = [None if r is None else float(torch.empty(1).uniform_(r[0], r[1])) for r in ranges]
    return order, fac


# ---------------------------------------------------------------------------------------------------- pixel operations
def rgb_to_l(img):
    """Pillow convert("L"): ITU-R 601-2 luma in 16.16 fixed point with rounding."""
    x = img.astype(np.uint32)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """Pillow Image.blend(deg, img, alpha): float32 arithmetic, truncation; clipped when alpha is outside [0, 1]."""
    a = np.float32(alpha)
    d = deg.astype(np.int32)
    t = d.astype(np.float32) + a * (img.astype(np.int32) - d).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    lum = rgb_to_l(img)
    mean = int(lum.astype(np.float64).sum() / lum.size + 0.5)
    return blend(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    lum = rgb_to_l(img)
    return blend(np.repeat(lum[..., None], 3, axis=-1), img, f)


def rgb_to_hsv(img):
    """Pillow convert("HSV") (Convert.c rgb2hsv_row): float32 ratios, float64 hue arithmetic, truncation."""
    r, g, b = (img[..., k].astype(np.int32) for k in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    gray = maxc == minc
    cr = np.where(gray, 1, maxc - minc).astype(np.float32)
    s = cr / np.where(gray, 1, maxc).astype(np.float32)
    rc = (maxc - r).astype(np.float32) / cr
    gc = (maxc - g).astype(np.float32) / cr
    bc = (maxc - b).astype(np.float32) / cr
    rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)
    h = np.where(r == maxc, (bc - gc).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    out = np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1)
    return out.astype(np.uint8)


def hsv_to_rgb(hsv):
    """Pillow HSV -> RGB (Convert.c hsv2rgb, after colorsys.py)."""
    h, s, v = (hsv[..., k].astype(np.int32) for k in range(3))
    h6 = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(h6).astype(np.int32)
    f = (h6 - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vf = v.astype(np.float32).astype(np.float64)

    def rnd(x):     # C round(): half away from zero (all values are >= 0 here)
        return np.clip(np.floor(x + 0.5).astype(np.int32), 0, 255)
    p = rnd(vf * (1.0 - fs))
    q = rnd(vf * (1.0 - fs * f))
    t = rnd(vf * (1.0 - fs * (1.0 - f)))
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v])
    g = np.choose(k, [t, v, v, q, p, p])
    b = np.choose(k, [p, p, t, v, v, q])
    gray = s == 0
    out = np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], -1)
    return out.astype(np.uint8)


def adjust_hue(img, f):
    if not (-0.5 <= f <= 0.5):
        raise ValueError("hue_factor ({}) is not in [-0.5, 0.5].".format(f))
    hsv = rgb_to_hsv(img)
    with np.errstate(over="ignore"):
        hsv[..., 0] += np.uint8(int(f * 255) & 0xff)      # np.uint8(hue_factor * 255): C-style truncation, wraps
    return hsv_to_rgb(hsv)


OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def color_jitter(img, order, factors):
    """img uint8 [h, w, 3]; applies the drawn operations in the drawn order."""
    out = np.ascontiguousarray(img, np.uint8)
    for fn_id in order:
        if factors[fn_id] is not None:
            out = OPS[fn_id](out, factors[fn_id])
    return out
