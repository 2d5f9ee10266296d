"""Oracle: torch-CPU restatement of the PMF training objective (TEST INFRASTRUCTURE).

focal      pc_processor/loss/focal_softmax.py:28-63   (softmax=False: input is probabilities)
lovasz     pc_processor/loss/lovasz_softmax.py:56-145 (classes='present', per_image=False, ignore=0)
perception tasks/pmf/trainer.py:231-252, entropy :305-319, total :330-332
"""
import math

import torch
import torch.nn.functional as F


def focal_loss(prob, target, alpha, gamma=2.0, mask=None):
    c = prob.shape[1]
    p = prob.permute(0, 2, 3, 1).reshape(-1, c)
    t = target.reshape(-1)
    pt = p.gather(1, t[:, None]).squeeze(1)
    loss = -(1 - pt).pow(gamma) * pt.clamp(1e-6).log() * alpha.to(prob.device, prob.dtype)[t]
    if mask is None:
        return loss.mean()
    m = mask.reshape(-1).to(loss.dtype)
    return (loss * m).sum() / m.sum()


def lovasz_softmax(prob, target, ignore=0):
    c = prob.shape[1]
    p = prob.permute(0, 2, 3, 1).reshape(-1, c)
    t = target.reshape(-1)
    valid = t != ignore
    p, t = p[valid], t[valid]
    if p.numel() == 0:
        return p * 0.
    terms = []
    for k in range(c):
        fg = (t == k).to(p.dtype)
        if fg.sum() == 0:
            continue
        err = (fg - p[:, k]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        fgs = fg[perm]
        total = fgs.sum()
        inter = total - fgs.cumsum(0)
        union = total + (1 - fgs).cumsum(0)
        jac = 1. - inter / union
        if jac.numel() > 1:
            jac = torch.cat((jac[:1], jac[1:] - jac[:-1]))
        terms.append(torch.dot(err_sorted, jac))
    return sum(terms) / len(terms)


def entropy_norm(prob):
    c = prob.shape[1]
    logp = torch.log(prob.clamp(min=1e-8))
    return -(prob * logp).sum(1) / math.log(c), logp


def perception_aware(pcd_prob, img_prob, tau=0.7):
    pe, plog = entropy_norm(pcd_prob)
    ie, ilog = entropy_norm(img_prob)
    pc, ic = 1 - pe, 1 - ie
    d = pc - ic
    w_pcd = d.gt(0).to(d.dtype) * d.abs() * pc.ge(tau).to(d.dtype)
    w_img = d.lt(0).to(d.dtype) * d.abs() * ic.ge(tau).to(d.dtype)
    kl = lambda logq, p: F.kl_div(logq, p, reduction="none")
    l_pcd = (kl(plog, img_prob) * w_img.unsqueeze(1)).mean()
    l_img = (kl(ilog, pcd_prob) * w_pcd.unsqueeze(1)).mean()
    return l_pcd + l_img


def pmf_total_loss(lidar_prob, camera_prob, label, alpha, lambda_=1.0, gamma_=0.5, tau=0.7):
    """tasks/pmf/trainer.py:303-332.  Returns (total, dict of the five terms)."""
    mask = label.gt(0)
    foc = focal_loss(lidar_prob, label, alpha, 2.0, mask)
    lov = lovasz_softmax(lidar_prob, label, 0)
    foc_c = focal_loss(camera_prob, label, alpha, 2.0, mask)
    lov_c = lovasz_softmax(camera_prob, label, 0)
    per = perception_aware(lidar_prob, camera_prob, tau)
    total = foc + lov * lambda_ + foc_c + lov_c * lambda_ + per * gamma_
    return total, {"foc": foc, "lov": lov, "foc_cam": foc_c, "lov_cam": lov_c, "per": per}
