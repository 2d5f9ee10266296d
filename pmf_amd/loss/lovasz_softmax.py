"""Lovasz-softmax -- call surface of pc_processor/loss/lovasz_softmax.py:71-160 (classes='present',
per_image=False, ignore=<label>), restated without data-dependent shapes.

The reference drops ignored pixels with boolean indexing (a device->host sync) and loops over classes with one
sort each (19 sorts per head).  Here ignored pixels keep their slot but get error -1 (they sort behind every valid
pixel, where they cannot change any valid prefix sum) and weight 0 in the final dot product; all classes are sorted
by ONE batched sort of a [C, P] matrix.  Per class the value is identical to the reference's:
    errors = |fg - p_c| sorted descending; jaccard = 1 - (G - cumsum(fg)) / (G + cumsum(1 - fg)); first differences;
    dot(errors, grad); mean over the classes present among the valid pixels."""
import ctypes as _C

import torch
import torch.nn as nn


def _jaccard_grad(fg_s, nv_s, n_valid):
    """first differences of the Jaccard index along each sorted class row.
    GPU: one HIP kernel pair (pmf_lovasz_grad) instead of two cumsums + six element-wise ops; CPU: torch ops."""
    if fg_s.is_cuda:
        from .. import _lib as L
        lib = L.lib()
        c, p = fg_s.shape
        fg_s = fg_s.contiguous()
        grad = torch.empty_like(fg_s)
        bsum = torch.empty((c, (p + 4095) // 4096), dtype=torch.float32, device=fg_s.device)
        nv = n_valid.reshape(1).to(torch.int64)
        rc = lib.pmf_lovasz_grad(fg_s.data_ptr(), c, p, nv.data_ptr(), bsum.data_ptr(), grad.data_ptr(),
                                 _C.c_void_p(torch.cuda.current_stream(fg_s.device).cuda_stream))
        L.check(rc, "pmf_lovasz_grad")
        return grad
    gts = fg_s.sum(1, keepdim=True)
    inter = gts - fg_s.cumsum(1)
    union = gts + ((1 - fg_s) * nv_s).cumsum(1)
    jac = 1. - inter / union.clamp_min(1e-12)       # union == 0 only in all-void slots (weight 0)
    return torch.cat((jac[:, :1], jac[:, 1:] - jac[:, :-1]), 1) * nv_s


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    if per_image:
        losses = [lovasz_softmax(p[None], l[None], classes, False, ignore) for p, l in zip(probas, labels)]
        return sum(losses) / max(len(losses), 1)
    if classes not in ("present", "all"):
        raise NotImplementedError("explicit class lists are not used by the PMF trainers")
    if probas.dim() == 3:
        probas = probas[:, None]
    c = probas.size(1)
    p = probas.permute(1, 0, 2, 3).reshape(c, -1)              # [C, P]
    lab = labels.reshape(-1)
    valid = torch.ones_like(lab, dtype=torch.bool) if ignore is None else (lab != ignore)
    vf = valid.to(p.dtype)
    cls = torch.arange(c, device=p.device)[:, None]
    fg = ((lab[None, :] == cls) & valid[None, :]).to(p.dtype)   # [C, P]
    err = (fg - p).abs()
    key = torch.where(valid[None, :], err.detach(), err.new_full((), -1.0))
    _, perm = torch.sort(key, dim=1, descending=True)
    nv_s = vf[perm]                                              # 1 for valid slots (they sort first)
    err_s = err.gather(1, perm) * nv_s                           # ignored pixels contribute 0
    fg_s = fg.gather(1, perm)
    grad = _jaccard_grad(fg_s, nv_s, valid.sum())                # constant w.r.t. the probabilities
    per_class = (err_s * grad).sum(1)
    if classes == "all":
        return per_class.mean()
    present = (fg.sum(1) > 0).to(p.dtype)
    return (per_class * present).sum() / present.sum().clamp_min(1.0)


class Lovasz_softmax(nn.Module):
    def __init__(self, classes="present", per_image=False, ignore=None):
        super().__init__()
        self.classes, self.per_image, self.ignore = classes, per_image, ignore

    def forward(self, probas, labels):
        return lovasz_softmax(probas, labels, self.classes, self.per_image, self.ignore)
