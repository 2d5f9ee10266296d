"""Focal loss on class probabilities -- call surface of pc_processor/loss/focal_softmax.py:7-63.

Device-agnostic torch ops (the tensors stay on the GPU; no host synchronisation).  Rank (f) of SURVEY.md 8
("loss stack as fused HIP kernels") is the next step for this module."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class FocalSoftmaxLoss(nn.Module):
    def __init__(self, n_classes, gamma=1, alpha=0.8, softmax=True):
        super().__init__()
        self.gamma, self.n_classes, self.softmax = gamma, n_classes, softmax
        if isinstance(alpha, list):
            assert len(alpha) == n_classes, "len(alpha)!=n_classes: {} vs. {}".format(len(alpha), n_classes)
            a = torch.Tensor(alpha)
        elif isinstance(alpha, np.ndarray):
            assert alpha.shape[0] == n_classes, "len(alpha)!=n_classes: {} vs. {}".format(alpha.shape[0], n_classes)
            a = torch.from_numpy(alpha)
        elif torch.is_tensor(alpha):
            a = alpha.detach().clone()
        else:
            assert 0 < alpha < 1, "invalid alpha: {}".format(alpha)
            a = torch.full((n_classes,), 1.0 - alpha)
            a[0] = alpha
        self.register_buffer("alpha", a.float(), persistent=False)

    def forward(self, x, target, mask=None):
        """x: [N,C,H,W] (or [P,C]) probabilities (softmax=False) or logits; target: [N,H,W] (or [P]) int64."""
        if x.dim() > 2:
            c = x.size(1)
            pred = x.reshape(x.size(0), c, -1).transpose(1, 2).reshape(-1, c)
        else:
            pred = x
        t = target.reshape(-1)
        if self.softmax:
            pred = F.softmax(pred, 1)
        pt = pred.gather(1, t[:, None]).squeeze(1)
        alpha = self.alpha.to(pt.device, pt.dtype)
        loss = -(1 - pt).pow(self.gamma) * pt.clamp(1e-6).log() * alpha[t]
        if mask is None:
            return loss.mean()
        m = mask.reshape(-1).to(loss.dtype)
        return (loss * m).sum() / m.sum()
