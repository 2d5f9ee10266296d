"""Perception-aware loss and the PMF training objective (tasks/pmf/trainer.py:231-252, 303-332)."""
import math

import torch
import torch.nn.functional as F


def normalized_entropy(prob):
    """E = -sum p log(clamp(p,1e-8)) / log(C)  and the log-probabilities (trainer.py:305-308)."""
    logp = torch.log(prob.clamp(min=1e-8))
    return -(prob * logp).sum(1) / math.log(prob.shape[1]), logp


def perception_aware_loss(pcd_prob, img_prob, tau=0.7, pcd_log=None, img_log=None, pcd_entropy=None,
                          img_entropy=None):
    if pcd_log is None:
        pcd_entropy, pcd_log = normalized_entropy(pcd_prob)
    if img_log is None:
        img_entropy, img_log = normalized_entropy(img_prob)
    pc, ic = 1 - pcd_entropy, 1 - img_entropy
    d = pc - ic
    w_pcd = d.gt(0).to(d.dtype) * d.abs() * pc.ge(tau).to(d.dtype)
    w_img = d.lt(0).to(d.dtype) * d.abs() * ic.ge(tau).to(d.dtype)
    l_pcd = (F.kl_div(pcd_log, img_prob, reduction="none") * w_img.unsqueeze(1)).mean()
    l_img = (F.kl_div(img_log, pcd_prob, reduction="none") * w_pcd.unsqueeze(1)).mean()
    return l_pcd + l_img, w_pcd, w_img


def pmf_total_loss(lidar_prob, camera_prob, label, focal, lovasz, lambda_=1.0, gamma_=0.5, tau=0.7):
    """total = foc + lambda*lov (both heads) + gamma*per; returns (total, dict of terms + entropies)."""
    mask = label.gt(0)
    pe, plog = normalized_entropy(lidar_prob)
    ie, ilog = normalized_entropy(camera_prob)
    t = {"foc": focal(lidar_prob, label, mask=mask), "lov": lovasz(lidar_prob, label),
         "foc_cam": focal(camera_prob, label, mask=mask), "lov_cam": lovasz(camera_prob, label)}
    t["per"], t["w_pcd"], t["w_img"] = perception_aware_loss(lidar_prob, camera_prob, tau, plog, ilog, pe, ie)
    t["pcd_entropy"], t["img_entropy"] = pe, ie
    total = t["foc"] + t["lov"] * lambda_ + t["foc_cam"] + t["lov_cam"] * lambda_ + t["per"] * gamma_
    return total, t


def epmf_total_loss(lidar_prob, camera_prob, label, focal, lovasz, mt_loss, tau=0.7):
    """the EPMF multi-task objective with torch ops (tasks/epmf/trainer.py:376-430, use_mtloss): the six terms
    [foc_img, lov_img, per_img, per, foc, lov] through MultiTaskLoss in THAT order (sigma index = list position).
    returns (total, dict of terms)."""
    mask = label.gt(0)
    pe, plog = normalized_entropy(lidar_prob)
    ie, ilog = normalized_entropy(camera_prob)
    pc, ic = 1 - pe, 1 - ie
    d = pc - ic
    w_pcd = d.gt(0).to(d.dtype) * d.abs() * pc.ge(tau).to(d.dtype)
    w_img = d.lt(0).to(d.dtype) * d.abs() * ic.ge(tau).to(d.dtype)
    per = (F.kl_div(plog, camera_prob, reduction="none") * w_img.unsqueeze(1)).mean()        # loss_per (pcd)
    per_img = (F.kl_div(ilog, lidar_prob, reduction="none") * w_pcd.unsqueeze(1)).mean()     # loss_per_img
    t = {"foc_cam": focal(camera_prob, label, mask=mask), "lov_cam": lovasz(camera_prob, label),
         "per_img": per_img, "per": per,
         "foc": focal(lidar_prob, label, mask=mask), "lov": lovasz(lidar_prob, label)}
    total = mt_loss([t[k].reshape(1) for k in EPMF_TERMS]).squeeze()
    return total, t


EPMF_TERMS = ("foc_cam", "lov_cam", "per_img", "per", "foc", "lov")      # the reference's loss_list order
