"""Oracle: numpy restatement of the SalsaNext range-image loader (TEST INFRASTRUCTURE -- never imported by pmf_amd).

Follows pc_processor/dataset/preprocess/projection.py:31-86 (RangeProjection.doProjection),
pc_processor/dataset/salsanext_loader.py:48-84 (feature / label / mask assembly, is_train=False or a pre-augmented
sweep) and pc_processor/dataset/preprocess/augmentor.py:97-180 (the rigid augmentation given its random draws).
Pinned against tests/golden/g10_range.npz, produced by running the reference classes (oracle/make_golden.py range_loader).

All projection arithmetic is float32, as in the reference (float32 arrays combined with Python scalars stay float32).
The reference picks the per-pixel winner by "argsort(depth)[::-1], then last writer wins" = the nearest point; numpy's
default argsort is not stable, so equal depths on one pixel are an open case there -- this restatement (and the HIP
kernel) resolve it to the smaller point index."""
import numpy as np


def fov_constants(fov_up, fov_down, fov_left=-180.0, fov_right=180.0):
    """projection.py:9-24 -> float32 (|fov_left|, fov_h, |fov_down|, fov_v) in radians."""
    assert fov_up >= 0 and fov_down <= 0 and fov_right >= 0 and fov_left <= 0
    up, down = fov_up / 180.0 * np.pi, fov_down / 180.0 * np.pi
    left, right = fov_left / 180.0 * np.pi, fov_right / 180.0 * np.pi
    return (np.float32(abs(left)), np.float32(abs(left) + abs(right)), np.float32(abs(down)),
            np.float32(abs(up) + abs(down)))


def pixel_coordinates(points, fov, proj_h, proj_w):
    """projection.py:33-58 -> (col int32[P], row int32[P], depth f32[P], col_f f32[P], row_f f32[P])."""
    p = np.asarray(points, np.float32)
    la, fh, da, fv = fov
    sq = p[:, :3] * p[:, :3]
    depth = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])
    yaw = -np.arctan2(p[:, 1], p[:, 0])
    pitch = np.arcsin(p[:, 2] / depth)
    cf = (yaw + la) / fh * np.float32(proj_w)
    rf = (np.float32(1.0) - (pitch + da) / fv) * np.float32(proj_h)
    col = np.clip(np.floor(cf), 0, proj_w - 1).astype(np.int32)
    row = np.clip(np.floor(rf), 0, proj_h - 1).astype(np.int32)
    return col, row, depth, cf, rf


def do_projection(points, fov, proj_h, proj_w):
    """projection.py:31-86 -> (proj_pointcloud f32[H,W,C], proj_range f32[H,W], proj_idx i32[H,W], proj_mask i32[H,W],
    uproj_x, uproj_y, uproj_depth)."""
    p = np.asarray(points, np.float32)
    col, row, depth, _, _ = pixel_coordinates(p, fov, proj_h, proj_w)
    pix = row.astype(np.int64) * proj_w + col
    order = np.lexsort((np.arange(p.shape[0]), depth, pix))       # by pixel, then depth, then index
    first = np.ones(order.shape[0], bool)
    first[1:] = pix[order][1:] != pix[order][:-1]
    win = order[first]                                            # nearest point of every occupied pixel
    idx = np.full(proj_h * proj_w, -1, np.int32)
    idx[pix[win]] = win
    idx = idx.reshape(proj_h, proj_w)
    hit = idx >= 0
    rng_img = np.full((proj_h, proj_w), -1, np.float32)
    rng_img[hit] = depth[idx[hit]]
    pc = np.full((proj_h, proj_w, p.shape[1]), -1, np.float32)
    pc[hit] = p[idx[hit]]
    return pc, rng_img, idx, (idx > 0).astype(np.int32), col, row, depth


def loader_item(points, mapped_label, fov, proj_h, proj_w, mean, stds):
    """salsanext_loader.py:53-84 -> (feature f32[5,H,W], label f32[H,W], mask i32[H,W], range f32[H,W], ux, uy, ud)."""
    pc, rng_img, idx, mask, ux, uy, ud = do_projection(points, fov, proj_h, proj_w)
    m = idx > 0
    label = np.zeros((proj_h, proj_w), np.float32)
    label[m] = np.asarray(mapped_label)[idx[m]]
    label = label * mask.astype(np.float32)
    inten = (pc[..., 3] != -1).astype(np.float32) * pc[..., 3]
    feat = np.concatenate([rng_img[None], pc[..., :3].transpose(2, 0, 1), inten[None]], 0)
    feat = (feat - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(stds, np.float32)[:, None, None]
    feat = feat * mask[None].astype(np.float32)
    return feat.astype(np.float32), label, mask, rng_img, ux, uy, ud


def euler_zyx_matrix(yaw, pitch, roll, degrees=True):
    """scipy Rotation.from_euler("zyx", [yaw, pitch, roll]) (extrinsic z, then y, then x) as a float64 matrix:
    R = Rx(roll) . Ry(pitch) . Rz(yaw)."""
    a, b, c = (np.deg2rad([yaw, pitch, roll]) if degrees else (yaw, pitch, roll))
    rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    return rx @ ry @ rz


def draw_augmentation(params, rnd):
    """augmentor.py:123-180: the order of the random draws (rnd: a random.Random-like object) ->
    (flipx, flipy, (tx,ty,tz), (roll,pitch,yaw))."""
    def maybe(p, lo, hi):
        return rnd.uniform(lo, hi) if rnd.uniform(0, 1) < p else 0
    flipx = rnd.uniform(0, 1) < params["p_flipx"]
    flipy = rnd.uniform(0, 1) < params["p_flipy"]
    t = tuple(maybe(params["p_trans" + a], params["trans_%smin" % a], params["trans_%smax" % a]) for a in "xyz")
    r = tuple(maybe(params["p_rot_" + a], params["rot_%smin" % a], params["rot_%smax" % a])
              for a in ("roll", "pitch", "yaw"))
    return flipx, flipy, t, r


def apply_augmentation(points, flipx, flipy, trans, rot_matrix):
    """augmentor.py:97-121 on a copy: flips, float32 translation, xyz <- float32(float64 xyz . R^T)."""
    p = np.array(points, np.float32, copy=True)
    if flipx:
        p[:, 0] = -p[:, 0]
    if flipy:
        p[:, 1] = -p[:, 1]
    for k in range(3):
        p[:, k] += trans[k]
    p[:, :3] = np.matmul(p[:, :3], np.asarray(rot_matrix, np.float64).T)
    return p
