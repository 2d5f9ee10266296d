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


def lidar_sweep(seed, npts, fov_up=3.0, fov_down=-25.0, duplicates=False):
    """A spinning-LiDAR-like sweep: f32[npts,4] (x,y,z,intensity), raw labels int32[npts], label LUT int32[260].
    Azimuth covers the full circle, elevation a little more than [fov_down, fov_up] (so rows clamp at both edges),
    ranges 1.5..80 m with ring structure; a few points sit on the coordinate axes and point 0 is a valid hit (the
    reference's mask drops it).  duplicates=True repeats a tenth of the points exactly (equal-depth ties)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    az = rng.uniform(-np.pi, np.pi, npts)
    el = np.deg2rad(rng.uniform(fov_down - 2.0, fov_up + 2.0, npts))
    r = 1.5 + 78.5 * rng.random(npts) ** 2
    x, y, z = r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)
    pts = np.stack([x, y, z, rng.random(npts)], 1).astype(np.float32)
    pts[0, :3] = (1.2, 0.05, -0.1)           # nearer than every other point: wins its pixel, and the mask drops it
    pts[1, :3] = (10.0, 0.0, 0.0)
    pts[2, :3] = (0.0, 7.5, 0.0)
    pts[3, :3] = (-12.0, 0.0, -1.0)          # yaw = -pi exactly
    pts[4, :3] = (0.0, -3.0, 0.5)
    if duplicates:
        k = npts // 10
        pts[k:2 * k] = pts[2 * k:3 * k]
    sem = rng.integers(0, 260, npts).astype(np.int32)
    lut = rng.integers(0, 20, 260).astype(np.int32)
    return pts, sem, lut


_AUG = dict(p_flipx=0., p_flipy=0.5, p_transx=0.5, trans_xmin=-5, trans_xmax=5, p_transy=0.5, trans_ymin=-3, trans_ymax=3,
            p_transz=0.5, trans_zmin=-1, trans_zmax=0., p_rot_roll=0.5, rot_rollmin=-5, rot_rollmax=5, p_rot_pitch=0.5,
            rot_pitchmin=-5, rot_pitchmax=5, p_rot_yaw=0.5, rot_yawmin=5, rot_yawmax=-5)
_MEAN, _STDS = [12.12, 10.88, 0.23, -1.04, 0.21], [12.32, 11.47, 6.91, 0.86, 0.16]
# (tag, seed, points, loader config) -- the sensor blocks of tasks/salsanext/config_server_{kitti,nus}.yaml
RANGE_CASES = (
    ("kitti", 0, 30000, {"augmentation": _AUG, "sensor": dict(proj_h=64, proj_w=512, fov_up=3., fov_down=-25., fov_left=-45,
                                                             fov_right=45, img_mean=_MEAN, img_stds=_STDS)}),
    ("nus", 3, 30000, {"augmentation": _AUG, "sensor": dict(proj_h=32, proj_w=2048, fov_up=10., fov_down=-30.,
                                                           fov_left=-180, fov_right=180, img_mean=_MEAN, img_stds=_STDS)}),
)


def kitti_tree(root, seed=0, seqs=(0, 8), frames=3, npts=500, h=24, w=80):
    """writes a small SemanticKITTI-layout tree (velodyne/labels/image_2/calib.txt per sequence) + its label config;
    returns (config_path, {(seq, frame): (points, raw uint32 labels, image)})."""
    import os
    import yaml
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = [0, 1, 10, 11, 13, 30, 40, 44, 48, 50, 70, 72, 80, 252, 259]
    learning = {k: (i % 6) for i, k in enumerate(ids)}
    inv = {v: k for k, v in sorted(learning.items(), reverse=True)}
    cfg = {"color_map": {k: [int(x) for x in rng.integers(0, 256, 3)] for k in ids},
           "color_map_inv": {k: [int(x) for x in rng.integers(0, 256, 3)] for k in inv},
           "learning_map": learning, "learning_map_inv": inv,
           "content": {k: float(rng.random()) for k in ids},
           "mapped_class_name": {k: "class%d" % k for k in inv}}
    os.makedirs(root, exist_ok=True)
    cfg_path = os.path.join(root, "labels.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    data = {}
    for s in seqs:
        sd = os.path.join(root, "%02d" % s)
        for sub in ("velodyne", "labels", "image_2"):
            os.makedirs(os.path.join(sd, sub), exist_ok=True)
        P2 = np.array([[0.58 * w, 0, w / 2.0, 4.5e1], [0, 0.58 * w, h / 2.0, -0.3], [0, 0, 1.0, 2.7e-3]]) + 1e-3 * s
        Tr = np.array([[4.2e-4, -9.9996e-1, -8.4e-3, -1.2e-2], [-7.2e-3, 8.4e-3, -9.9993e-1, -5.4e-2],
                       [9.9997e-1, 4.8e-4, -7.2e-3, -2.9e-1]])
        with open(os.path.join(sd, "calib.txt"), "w") as f:
            f.write("P0: " + " ".join("%.12e" % v for v in np.zeros(12)) + "\n")
            f.write("P2: " + " ".join("%.12e" % v for v in P2.reshape(-1)) + "\n")
            f.write("Tr: " + " ".join("%.12e" % v for v in Tr.reshape(-1)) + "\n")
        for fr in range(frames):
            n = npts + 7 * fr
            pts = np.stack([rng.uniform(-5, 60, n), rng.uniform(-25, 25, n), rng.uniform(-3, 2.5, n), rng.random(n)],
                           1).astype(np.float32)
            sem = rng.choice(ids, n).astype(np.uint32)
            inst = rng.integers(0, 300, n).astype(np.uint32)
            raw = (inst << 16) | sem
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            pts.tofile(os.path.join(sd, "velodyne", "%06d.bin" % fr))
            raw.tofile(os.path.join(sd, "labels", "%06d.label" % fr))
            Image.fromarray(img).save(os.path.join(sd, "image_2", "%06d.png" % fr))
            data[("%02d" % s, "%06d" % fr)] = (pts, raw, img)
    return cfg_path, data


def merge_case(seed, pc_size, n_cams=6, ncls=17):
    """per-camera (point_idx int64, conf f32, label int64) lists for the nuScenes view merge: ~30 % of the points per
    view, unseen points, exact confidence ties between views, and zero-confidence entries (padded image rows)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    idx, conf, lab = [], [], []
    for j in range(n_cams):
        sel = np.sort(rng.choice(pc_size, int(0.3 * pc_size), replace=False)).astype(np.int64)
        sel = sel[sel >= 8]                                   # points 0..7 are placed by hand below
        c = rng.uniform(0.06, 1.0, sel.size).astype(np.float32)
        c = (np.round(c * 64) / 64).astype(np.float32)       # coarse values: ties between views are frequent
        l = rng.integers(0, ncls, sel.size).astype(np.int64)
        idx.append(sel); conf.append(c); lab.append(l)

    def put(j, p, c, l):
        idx[j] = np.append(idx[j], np.int64(p)); conf[j] = np.append(conf[j], np.float32(c)); lab[j] = np.append(lab[j], np.int64(l))
    put(2, 1, 0.0, 0)                  # only a zero-confidence entry, not in view 0 -> -1
    put(0, 2, 0.0, 5); put(3, 2, 0.0, 7)   # zero confidence in view 0 too -> view 0's label
    put(1, 3, 0.5, 4); put(4, 3, 0.5, 9)   # tie -> first view
    put(5, 4, 0.25, 0)                 # label 0 from the last view
    put(0, 5, 0.3, 2); put(1, 5, 0.9, 11)  # higher confidence wins
    return idx, conf, lab              # points 0, 6, 7 unseen


class SyntheticNus(object):
    """A devkit-free stand-in for pc_processor/dataset/nuScenes/dataset_nuscenes.py:74-282 with the attributes
    NusPerspectiveViewLoader (tasks/pmf_eval_nuscenes/nus_perspective_loader.py) and the inference loop
    (tasks/pmf_eval_nuscenes/infer.py:110-200) use: six consecutive indices = the six camera views of ONE sweep
    (token_list[i] = {lidar_token, cam_token}), loadDataByIndex -> (f32[P,4] x/y/z/intensity, uint8[P,1] raw labels, int32[P]),
    loadImage -> uint8[h,w,3], labelMapping (the vectorised dictionary of dataset_nuscenes.py:181-186) and
    mapLidar2Camera(index, xyz, img_h=<image width>, img_w=<image height>) -> ((row, col) float of the kept points, keep
    mask): same masks as :268-276 (depth > 1 m, one pixel margin), the devkit's chain of four rigid transforms replaced by
    ONE yaw rotation per camera (cameras 60 degrees apart, 80-degree horizontal field of view: neighbouring views overlap,
    the merge has to decide) and a pinhole model.  TEST INFRASTRUCTURE: the reference loader class is executed against
    this object to produce tests/golden/g15_nus.npz, the HIP loader and the numpy oracle are compared on it."""

    N_CAM = 6

    def __init__(self, seed=0, sweeps=2, npts=6000, h=80, w=160, nclasses=17):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.h, self.w, self.nclasses = h, w, nclasses
        self.sweeps, self.images = [], []
        for s in range(sweeps):
            pts, _, _ = lidar_sweep(100 * seed + s, npts)
            raw = rng.integers(0, 32, (npts, 1)).astype(np.uint8)
            self.sweeps.append((pts, raw))
            self.images.append([rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(self.N_CAM)])
        # general index (0..31) -> segmentation index (0..16), as map_name_from_general_index_to_segmentation_index
        self.map_name_from_general_index_to_segmentation_index = {i: int(rng.integers(0, nclasses)) for i in range(32)}
        self.mapped_cls_name = {i: "class_%d" % i for i in range(nclasses)}
        self.token_list = [{"lidar_token": "sweep%03d" % (i // self.N_CAM), "cam_token": "sweep%03d_cam%d" % (
            i // self.N_CAM, i % self.N_CAM)} for i in range(sweeps * self.N_CAM)]
        self.fx = 0.6 * w

    def __len__(self):
        return len(self.token_list)

    def parsePathInfoByIndex(self, index):
        return index, ""

    def loadDataByIndex(self, index):
        pts, raw = self.sweeps[index // self.N_CAM]
        return pts, raw, np.zeros(pts.shape[0], dtype=np.int32)

    def loadImage(self, index):
        return self.images[index // self.N_CAM][index % self.N_CAM]

    def labelMapping(self, sem_label):
        sem_label = np.vectorize(self.map_name_from_general_index_to_segmentation_index.__getitem__)(sem_label)
        assert sem_label.shape[-1] == 1
        return sem_label[:, 0]

    def mapLidar2Camera(self, index, pointcloud, img_h, img_w, min_dist=1.0):
        yaw = np.deg2rad(60.0 * (index % self.N_CAM))
        x, y, z = (pointcloud[:, k].astype(np.float64) for k in range(3))
        fwd = np.cos(yaw) * x + np.sin(yaw) * y          # camera z (depth)
        right = np.sin(yaw) * x - np.cos(yaw) * y        # camera x
        down = -z - 0.3                                  # camera y
        with np.errstate(divide="ignore", invalid="ignore"):
            u = self.fx * right / fwd + 0.5 * self.w
            v = self.fx * down / fwd + 0.5 * self.h
        mask = np.ones(fwd.shape[0], dtype=bool)
        mask = np.logical_and(mask, fwd > min_dist)
        mask = np.logical_and(mask, u > 1)
        mask = np.logical_and(mask, u < img_h - 1)       # (the reference passes the image WIDTH as img_h, :272-273)
        mask = np.logical_and(mask, v > 1)
        mask = np.logical_and(mask, v < img_w - 1)
        mapped = np.stack([v, u], 1)                     # fliplr: (row, col)
        return mapped[mask, :], mask
