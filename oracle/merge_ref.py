"""Oracle: numpy restatement of the multi-camera prediction merge (TEST INFRASTRUCTURE).

Follows tasks/pmf_eval_nuscenes/infer.py:18-38 (getMergePred): confidence and label tables [n_cams, P] (0 / -1 where a
view does not see the point), argmax over the views (first maximum), the label of that view.  Pinned against
tests/golden/g12_merge.npz, produced by executing the reference's own function here (oracle/make_golden.py merge)."""
import numpy as np


def get_merge_pred(point_idx_list, pred_conf_list, pred_argmax_list, pc_size):
    n = len(point_idx_list)
    conf = np.zeros((n, pc_size), np.float32)
    lab = np.full((n, pc_size), -1, np.int64)
    for j in range(n):
        conf[j, point_idx_list[j]] = pred_conf_list[j]
        lab[j, point_idx_list[j]] = pred_argmax_list[j]
    best = conf.argmax(axis=0)                       # first maximum, as torch.argmax
    return lab[best, np.arange(pc_size)]
