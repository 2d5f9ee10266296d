"""Seeded synthetic input recipes shared by make_golden.py, tests and bench.py (TEST INFRASTRUCTURE).

Inputs are regenerated from these recipes wherever they are needed; only the expected outputs
of the reference are stored under tests/golden/."""
import numpy as np


def knn_case(seed, h, w, npts, quantize=False):
    """SURVEY 8(d) config-2 style KNN input: range image with ~30% empty (-1) pixels, points on
    occupied pixels (with repeats, incl. border pixels), per-point range = pixel range + U(0, 0.2).

    quantize=True rounds ranges to 0.25 m so equal distances (selection ties) are frequent."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = rng.uniform(2.0, 60.0, (h, w))
    smooth = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 1)) / 4.0
    pr = np.where(rng.random((h, w)) < 0.5, smooth, base).astype(np.float32)
    if quantize:
        pr = (np.round(pr * 4) / 4).astype(np.float32)
    empty = rng.random((h, w)) < 0.3
    pr[empty] = -1.0
    am = rng.integers(0, 20, (h, w)).astype(np.int64)
    am[empty] = 0
    occ = np.argwhere(~empty)
    sel = rng.integers(0, occ.shape[0], npts)
    sel[:64] = np.arange(64) % occ.shape[0]
    py = occ[sel, 0].astype(np.int64)
    px = occ[sel, 1].astype(np.int64)
    # force some border / corner points
    edge = np.argwhere(~empty[0])[:8, 0]
    if edge.size:
        py[64:64 + edge.size] = 0
        px[64:64 + edge.size] = edge
    jitter = rng.uniform(0, 0.2, npts)
    if quantize:
        jitter = np.round(jitter * 4) / 4
    ur = (pr[py, px] + jitter).astype(np.float32)
    return pr, ur, am, px, py
