"""Merge of the six camera views of one nuScenes sweep -- tasks/pmf_eval_nuscenes/infer.py:18-38 (getMergePred).

Each camera view yields, for the points it sees, a confidence (max class probability at the point's pixel) and a
label (after KNN).  Per LiDAR point the label of the most confident view wins; unseen points get -1.  The reference
scatters into a [6, P] table and then walks the P points in a Python loop; here it is three small HIP launches
(pmf_merge_pred: init, one atomicMax scatter per view, decode)."""
import ctypes as C

import torch

from .. import _lib as L


def getMergePred(point_idx_list, pred_conf_list, pred_argmax_list, pc_size, fallback=None):
    """-> int64 [pc_size] on the inputs' device (-1 = no view decides; same values as the reference).
    fallback (optional, int64 [pc_size]): labels of a LiDAR-only model (SalsaNext) taken for the points outside every
    camera view instead of -1 -- the test-split protocol of more_experiment_config.md:10, in the same launch."""
    n = len(point_idx_list)
    if not (n == len(pred_conf_list) == len(pred_argmax_list)) or n == 0:
        raise ValueError("getMergePred: need the same, non-zero number of index / confidence / label lists")
    dev = pred_conf_list[0].device
    if dev.type != "cuda":
        raise RuntimeError("pmf_amd getMergePred runs on the GPU only (no CPU fallback)")
    idx = [t.to(dev).long().contiguous() for t in point_idx_list]
    conf = [t.to(dev).float().contiguous() for t in pred_conf_list]
    lab = [t.to(dev).long().contiguous() for t in pred_argmax_list]
    for a, b, c in zip(idx, conf, lab):
        if not (a.numel() == b.numel() == c.numel()):
            raise ValueError("getMergePred: index / confidence / label lengths differ within a view")
    pc_size = int(pc_size)
    keys = torch.empty(max(pc_size, 1), dtype=torch.int64, device=dev)
    out = torch.empty(pc_size, dtype=torch.int64, device=dev)
    ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    counts = (C.c_int64 * n)(*[t.numel() for t in idx])
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if fallback is not None:
        fb = fallback.to(dev).long().contiguous()
        if fb.numel() != pc_size:
            raise ValueError("getMergePred: fallback needs one label per point (%d), got %d" % (pc_size, fb.numel()))
        L.check(L.lib().pmf_merge_pred_fallback(n, ptrs(idx), ptrs(conf), ptrs(lab), counts, pc_size, fb.data_ptr(),
                                                keys.data_ptr(), out.data_ptr(), st), "pmf_merge_pred_fallback")
        return out
    L.check(L.lib().pmf_merge_pred(n, ptrs(idx), ptrs(conf), ptrs(lab), counts, pc_size, keys.data_ptr(), out.data_ptr(),
                                   st), "pmf_merge_pred")
    return out
