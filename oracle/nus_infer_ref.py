"""Oracle: numpy / torch-CPU restatement of the nuScenes six-camera inference path (TEST INFRASTRUCTURE).

  nus_loader_item   tasks/pmf_eval_nuscenes/nus_perspective_loader.py:11-79  (per-view projection frame [8,h,w], mask, label,
                    row / column indices, depth and sweep indices of the kept points, sweep size)
  infer_sweeps      tasks/pmf_eval_nuscenes/infer.py:110-200  (per view: crop the top rows, normalise * mask, PMFNet, zero-pad
                    back, confidence / argmax, KNN or pixel lookup; after six views getMergePred; -1 -> 0, int32 labels per
                    sweep), with the LiDAR-only fallback of more_experiment_config.md:10 as an option

Pinned against tests/golden/g15_nus.npz: the reference's own NusPerspectiveViewLoader class and getMergePred function
executed here on oracle.cases.SyntheticNus (oracle/make_golden.py nus)."""
import numpy as np
import torch

from . import knn_ref, merge_ref


def nus_loader_item(dataset, index):
    pointcloud, sem_label, _ = dataset.loadDataByIndex(index)
    image = np.array(dataset.loadImage(index))
    seq_id, _ = dataset.parsePathInfoByIndex(index)
    mapped, keep = dataset.mapLidar2Camera(seq_id, pointcloud[:, :3], image.shape[1], image.shape[0])
    y_data = mapped[:, 1].astype(np.int32)
    x_data = mapped[:, 0].astype(np.int32)
    h, w = image.shape[:2]
    img = image.astype(np.float32) / 255.0
    depth = np.linalg.norm(pointcloud[:, :3], 2, axis=1)
    kept = pointcloud[keep]
    proj = np.zeros((10, h, w), np.float32)
    # numpy fancy assignment with repeated indices: the LAST kept point in file order wins
    lab = dataset.labelMapping(sem_label[keep])
    for k in range(kept.shape[0]):
        r, c = x_data[k], y_data[k]
        proj[0, r, c] = depth[keep][k]
        proj[1:5, r, c] = kept[k]
        proj[8, r, c] = 1.0
        proj[9, r, c] = float(lab[k])
    proj[5:8] = img.transpose(2, 0, 1)
    return (proj[:8], proj[8], proj[9], x_data, y_data, depth[keep].astype(np.float32),
            np.arange(pointcloud.shape[0])[keep], pointcloud.shape[0])


def infer_sweeps(dataset, predict, proj_h, mean, std, knn_params=None, nclasses=17, fallback=None):
    """predict(pcd[1,5,h,w], rgb[1,3,h,w]) -> probabilities [1,C,h,w] (torch, CPU); returns {lidar_token: int32[P]}.
    fallback(index) -> int64[P] labels of a LiDAR-only model for the sweep of view `index` (optional)."""
    out = {}
    idx_l, conf_l, lab_l = [], [], []
    mean = torch.tensor(mean).view(1, -1, 1, 1)
    std = torch.tensor(std).view(1, -1, 1, 1)
    for i in range(len(dataset)):
        feat, mask, _, ux, uy, udepth, pidx, psize = nus_loader_item(dataset, i)
        feat = torch.from_numpy(feat)[None]
        mask = torch.from_numpy(mask)[None]
        proj_depth = feat[0, 0].clone()
        proj_depth = proj_depth - proj_depth.eq(0).float()
        h_pad = feat.shape[2] - proj_h
        x = feat[:, :, h_pad:, :].clone()
        m = mask[:, h_pad:, :]
        x[:, 0:5] = (x[:, 0:5] - mean) / std * m.unsqueeze(1)
        pred = predict(x[:, 0:5], x[:, 5:8])
        pred = torch.nn.functional.pad(pred, (0, 0, h_pad, 0))
        conf, am = pred[0].max(dim=0)
        ux_t, uy_t = torch.from_numpy(ux).long(), torch.from_numpy(uy).long()
        if knn_params is not None:
            lab = torch.from_numpy(knn_ref.knn_vote(proj_depth.numpy(), udepth, am.numpy(), uy.astype(np.int64),
                                                    ux.astype(np.int64), **knn_params, nclasses=nclasses))
        else:
            lab = am[ux_t, uy_t]
        idx_l.append(pidx)
        conf_l.append(conf[ux_t, uy_t].numpy())
        lab_l.append(lab.numpy())
        if len(idx_l) == 6:
            merged = merge_ref.get_merge_pred(idx_l, conf_l, lab_l, psize)
            if fallback is not None:
                fb = np.asarray(fallback(i))
                merged = np.where(merged < 0, fb, merged)
            merged = merged * (merged != -1)
            out[dataset.token_list[i]["lidar_token"]] = merged.reshape(-1).astype(np.int32)
            idx_l, conf_l, lab_l = [], [], []
    return out
