"""Oracle: numpy restatement of pc_processor/postproc/knn.py:12-143 (TEST INFRASTRUCTURE).

Integer output (voted labels) must be bit-exact.  All distance arithmetic is
float32 in the same operation order as the reference (|neigh - range| * (1 - G)).

Tie rule.  The reference selects neighbours with ``topk(k, largest=False,
sorted=False)`` (knn.py:111-112) whose choice among EQUAL distances is
implementation-defined.  This oracle -- and the HIP kernel -- break ties by the
smaller tap index (row-major position in the search window).  ``boundary_tie``
flags the points whose k-th and (k+1)-th smallest distances are equal AND
finite; only on those may the reference legitimately differ.
"""
import math

import numpy as np


def gaussian_window(search: int, sigma: float) -> np.ndarray:
    """knn.py:12-34 -- normalised 2-D Gaussian.  Evaluated with torch float32 ops in the
    reference's operation order so the 25 weights are bit-identical to the reference's
    (numpy's and torch's float32 exp may differ in the last ulp)."""
    import torch
    c = torch.arange(search)
    xg = c.repeat(search).view(search, search)
    xy = torch.stack([xg, xg.t()], dim=-1).float()
    mean = (search - 1) / 2.
    var = sigma ** 2.
    g = (1. / (2. * math.pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2., dim=-1) / (2 * var))
    g = g / torch.sum(g)
    return g.numpy().astype(np.float32)


def knn_vote(proj_range, unproj_range, proj_argmax, px, py, knn=5, search=5, sigma=1.0,
             cutoff=1.0, nclasses=20, return_aux=False):
    """proj_range f32[H,W] (-1 = empty), unproj_range f32[P], proj_argmax i64[H,W],
    px (col) i64[P], py (row) i64[P]  ->  labels i64[P] in [1, nclasses-1]."""
    if search % 2 == 0:
        raise ValueError("Nearest neighbor kernel must be odd number")   # knn.py:73-74
    proj_range = np.asarray(proj_range, np.float32)
    unproj_range = np.asarray(unproj_range, np.float32)
    proj_argmax = np.asarray(proj_argmax, np.int64)
    px = np.asarray(px, np.int64)
    py = np.asarray(py, np.int64)
    H, W = proj_range.shape
    P = unproj_range.shape[0]
    assert px.shape[0] == P and py.shape[0] == P, "len(unproj_range) must equal len(px) == len(py)"
    pad = (search - 1) // 2
    S2 = search * search
    # F.unfold with zero padding (knn.py:80-82, 114-117)
    rp = np.zeros((H + 2 * pad, W + 2 * pad), np.float32)
    rp[pad:pad + H, pad:pad + W] = proj_range
    lp = np.zeros((H + 2 * pad, W + 2 * pad), np.int64)
    lp[pad:pad + H, pad:pad + W] = proj_argmax
    dy, dx = np.divmod(np.arange(S2), search)
    rows = py[:, None] + dy[None, :]
    cols = px[:, None] + dx[None, :]
    neigh = rp[rows, cols].copy()                 # [P, S2]
    labs = lp[rows, cols]
    neigh[neigh < 0] = np.inf                     # knn.py:91
    center = (S2 - 1) // 2
    neigh[:, center] = unproj_range               # knn.py:94-95
    with np.errstate(invalid="ignore"):
        dist = np.abs(neigh - unproj_range[:, None]).astype(np.float32)
    w = (np.float32(1) - gaussian_window(search, sigma)).reshape(1, S2)
    dist = (dist * w).astype(np.float32)          # knn.py:103-108
    order = np.argsort(dist, axis=1, kind="stable")
    sel = order[:, :knn]
    sd = np.take_along_axis(dist, sel, 1)
    sl = np.take_along_axis(labs, sel, 1).copy()
    if cutoff > 0:
        sl[sd > np.float32(cutoff)] = nclasses     # knn.py:124-127
    votes = np.zeros((P, nclasses + 1), np.int32)
    np.add.at(votes, (np.repeat(np.arange(P), knn), sl.reshape(-1)), 1)
    out = votes[:, 1:-1].argmax(1) + 1            # first max wins (knn.py:138)
    if not return_aux:
        return out.astype(np.int64)
    sdist = np.take_along_axis(dist, order, 1)
    kth, nxt = sdist[:, knn - 1], sdist[:, knn]
    boundary_tie = (kth == nxt) & np.isfinite(kth)
    return out.astype(np.int64), {"boundary_tie": boundary_tie, "sel": np.sort(sel, 1)}
