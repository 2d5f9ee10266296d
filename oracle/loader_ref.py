"""Oracle: numpy restatement of the perspective-projection loader (TEST INFRASTRUCTURE).

Follows pc_processor/dataset/semantic_kitti/parser.py:209-227 (mapLidar2Camera) and
pc_processor/dataset/perspective_view_loader.py:77-135 (scatter + stack) for the
deterministic ``return_uproj=True`` / validation path.  float64 projection,
int32 truncation, numpy fancy-index scatter (last point in file order wins).
"""
import numpy as np


def map_lidar_to_camera(proj_matrix, xyz, img_w, img_h):
    """parser.py:209-227.  proj_matrix f64[3,4] (= P2 @ Tr), xyz f32[P,3].

    The reference's parameter names are swapped at the call site
    (perspective_view_loader.py:89-90 passes image width first); here they carry
    their true meaning.  Returns (rowcol f64[K,2], keep bool[P])."""
    xyz = np.asarray(xyz, np.float32)
    keep = xyz[:, 0] > 0.5
    hom = np.concatenate([xyz[keep], np.ones((int(keep.sum()), 1), np.float32)], axis=1)
    m = (np.asarray(proj_matrix, np.float64) @ hom.T).T
    uv = m[:, :2] / m[:, 2:3]
    inside = (uv[:, 0] > 0) * (uv[:, 0] < img_w) * (uv[:, 1] > 0) * (uv[:, 1] < img_h)
    keep[keep] = inside
    return np.fliplr(uv)[inside], keep


def project_frame(points, sem_label, image_u8, proj_matrix, label_lut):
    """perspective_view_loader.py:79-135 -> (proj[10,h,w] f32, x_data i32[K], y_data i32[K], depth f32[P])."""
    points = np.asarray(points, np.float32)
    h, w = image_u8.shape[0], image_u8.shape[1]
    rowcol, keep = map_lidar_to_camera(proj_matrix, points[:, :3], w, h)
    x_data = rowcol[:, 0].astype(np.int32)
    y_data = rowcol[:, 1].astype(np.int32)
    img = image_u8.astype(np.float32) / 255.0
    depth = np.linalg.norm(points[:, :3], 2, axis=1)
    kept = points[keep]
    xyzi = np.zeros((h, w, 4), np.float32)
    xyzi[x_data, y_data] = kept
    pdepth = np.zeros((h, w), np.float32)
    pdepth[x_data, y_data] = depth[keep]
    plabel = np.zeros((h, w), np.int32)
    plabel[x_data, y_data] = np.asarray(label_lut)[np.asarray(sem_label)[keep]]
    pmask = np.zeros((h, w), np.int32)
    pmask[x_data, y_data] = 1
    proj = np.concatenate([pdepth[None], xyzi.transpose(2, 0, 1), img.transpose(2, 0, 1),
                           pmask[None].astype(np.float32), plabel[None].astype(np.float32)], 0)
    return proj.astype(np.float32), x_data, y_data, depth.astype(np.float32)


def center_crop_pad(proj, out_h, out_w, h_pad, w_pad):
    """Validation path of perspective_view_loader.py:71-74,138-141: CenterCrop((H-2hp, W-2wp)) then Pad((wp, hp)).

    torchvision CenterCrop semantics: if the image is smaller than the crop it is zero-padded
    symmetrically first (floor on the leading side); crop offset = round((size - crop) / 2)."""
    c, h, w = proj.shape
    ch, cw = out_h - 2 * h_pad, out_w - 2 * w_pad
    if cw > w or ch > h:
        pl = (cw - w) // 2 if cw > w else 0
        pt = (ch - h) // 2 if ch > h else 0
        pr = (cw - w + 1) // 2 if cw > w else 0
        pb = (ch - h + 1) // 2 if ch > h else 0
        proj = np.pad(proj, ((0, 0), (pt, pb), (pl, pr)))
        c, h, w = proj.shape
    top = int(round((h - ch) / 2.0))
    left = int(round((w - cw) / 2.0))
    out = proj[:, top:top + ch, left:left + cw]
    return np.pad(out, ((0, 0), (h_pad, h_pad), (w_pad, w_pad)))


def synthetic_frame(seed=0, n_points=5000, h=96, w=320):
    """A synthetic KITTI-like frame: calib, points (incl. duplicates per pixel, behind-camera,
    x<=0.5, out-of-frustum, near-integer u/v), labels, RGB image, label LUT."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f = 0.58 * w
    P2 = np.array([[f, 0, w / 2.0, 4.5e1], [0, f, h / 2.0, -0.3], [0, 0, 1.0, 2.7e-3]], np.float64)
    Tr = np.eye(4)
    Tr[:3, :4] = np.array([[4.2e-4, -9.9996e-1, -8.4e-3, -1.2e-2],
                           [-7.2e-3, 8.4e-3, -9.9993e-1, -5.4e-2],
                           [9.9997e-1, 4.8e-4, -7.2e-3, -2.9e-1]], np.float64)
    M = P2 @ Tr
    x = rng.uniform(-5, 60, n_points)
    y = rng.uniform(-25, 25, n_points)
    z = rng.uniform(-3, 2.5, n_points)
    pts = np.stack([x, y, z, rng.random(n_points)], 1).astype(np.float32)
    pts[: n_points // 10] = pts[n_points // 10: 2 * (n_points // 10)]          # exact duplicates
    pts[5, 0] = 0.5
    pts[6, 0] = np.float32(0.50000006)
    sem = rng.integers(0, 260, n_points).astype(np.int32)
    lut = np.zeros(260, np.int32)
    lut[:] = rng.integers(0, 20, 260)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    return M, pts, sem, img, lut
