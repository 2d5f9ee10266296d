"""Oracle: numpy restatement of the EPMF loader (TEST INFRASTRUCTURE).

Follows pc_processor/dataset/semantic_kitti/parser.py:229-257 (mapLidar2CameraCropYaw) and
pc_processor/dataset/perspective_view_loader_v2.py:42-157 for the deterministic paths (is_train=False;
``return_uproj=True`` is pinned against tests/golden/g9_loader_v2.npz, produced by running the reference class).
The final Pad + CenterCrop of the validation path are torchvision transforms (third party, absent here): restated
from torchvision's documented semantics, unpinned."""
import numpy as np

FOV_LEFT, FOV_RIGHT = -45 / 180.0 * np.pi, 45 / 180.0 * np.pi     # parser.py:36-37


def map_lidar_to_camera_crop_yaw(proj_matrix, pointcloud, fov_left=FOV_LEFT, fov_right=FOV_RIGHT):
    """parser.py:229-257 -> (crop_pointcloud f32[K,4], rowcol f64[K,2], keep bool[P])."""
    pointcloud = np.asarray(pointcloud, np.float32)
    depth = np.linalg.norm(pointcloud[:, :3], 2, axis=1)
    yaw = -np.arctan2(pointcloud[:, 1], pointcloud[:, 0])
    keep = np.logical_and(depth > 0.5, (yaw >= fov_left) * (yaw <= fov_right))
    crop = pointcloud[keep]
    hom = np.concatenate([crop[:, :3], np.ones([int(keep.sum()), 1], np.float32)], axis=1)
    m = (np.asarray(proj_matrix, np.float64) @ hom.T).T
    uv = m[:, :2] / np.expand_dims(m[:, 2], axis=1)
    return crop, np.fliplr(uv), keep


def project_frame_v2(points, sem_label, image_u8, proj_matrix, label_lut, img_scale=1.0):
    """perspective_view_loader_v2.py:57-140 -> (proj f32[10,h,w], xy_index f64[K,2], depth f32[K], keep bool[P])."""
    image = np.asarray(image_u8).astype(np.float32) / 255.0
    crop, xy_index, keep = map_lidar_to_camera_crop_yaw(proj_matrix, points)
    xy_index = xy_index * img_scale
    sem = np.asarray(sem_label)[keep]
    x_data = xy_index[:, 0].astype(np.int32)
    y_data = xy_index[:, 1].astype(np.int32)
    x_min, x_max, y_min, y_max = x_data.min(), x_data.max(), y_data.min(), y_data.max()
    h, w = x_max - x_min + 1, y_max - y_min + 1
    xyzi = np.zeros((h, w, crop.shape[1]), np.float32)
    xyzi[x_data - x_min, y_data - y_min] = crop
    pdepth = np.zeros((h, w), np.float32)
    depth = np.linalg.norm(crop[:, :3], 2, axis=1)
    pdepth[x_data - x_min, y_data - y_min] = depth
    plabel = np.zeros((h, w), np.int32)
    plabel[x_data - x_min, y_data - y_min] = np.asarray(label_lut)[sem]
    pmask = np.zeros((h, w), np.int32)
    pmask[x_data - x_min, y_data - y_min] = 1
    rgb = np.zeros((h, w, 3), np.float32)
    rr, cc = np.arange(h)[:, None] + x_min, np.arange(w)[None, :] + y_min       # :105-125 as one window test
    ok = (rr >= 0) & (rr < image.shape[0]) & (cc >= 0) & (cc < image.shape[1])
    rgb[ok] = image[np.broadcast_to(rr, ok.shape)[ok], np.broadcast_to(cc, ok.shape)[ok]]
    proj = np.concatenate([pdepth[None], xyzi.transpose(2, 0, 1), rgb.transpose(2, 0, 1),
                           pmask[None].astype(np.float32), plabel[None].astype(np.float32)], 0)
    return proj.astype(np.float32), xy_index, depth.astype(np.float32), keep


def pad_center_crop(proj, max_h, max_w, crop_h, crop_w):
    """:142-153 validation path: Pad((left, 0, right, bottom)) to (max(max_h,h), max(max_w,w)) then CenterCrop."""
    c, h, w = proj.shape
    mh, mw = max(max_h, h), max(max_w, w)
    left = (mw - w) // 2
    out = np.zeros((c, mh, mw), np.float32)
    out[:, :h, left:left + w] = proj
    top = int(round((mh - crop_h) / 2.0))
    lft = int(round((mw - crop_w) / 2.0))
    return out[:, top:top + crop_h, lft:lft + crop_w]
