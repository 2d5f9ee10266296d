"""The PMF training objective as ONE autograd node on the GPU (value + analytic gradient in HIP, include/pmf_amd.h
``pmf_loss_pixel`` / ``pmf_loss_lovasz``).

    total = foc + foc_cam + lambda * (lov + lov_cam) + gamma * per          (tasks/pmf/trainer.py:303-332)

``pmf_total_loss`` (perception.py) is the same objective written with torch ops, term by term like the reference; it
is what the CPU host tests and the float64 parity tests differentiate with autograd.  This module is what the
training engine runs on the GPU: ~250 element-wise launches per iteration become 6 kernels + one batched sort, and the
confusion matrices of both heads (argmax vs label) are updated in the same pass."""
import ctypes as C

import torch

from .. import _lib as L


_SORT_WS = {}


def _sort_workspace(lib, c, p, dev):
    """scratch of the in-library Lovasz sort (key / pixel ping-pong buffers + histograms), kept per shape and device"""
    k = (c, p, str(dev))
    ws = _SORT_WS.get(k)
    if ws is None:
        ws = _SORT_WS[k] = torch.empty(lib.pmf_loss_sort_workspace(c, p), dtype=torch.uint8, device=dev)
    return ws


# set by TrainEngine.train_step around its own `total.backward()` (upstream gradient exactly 1): the backward of the fused
# objective then hands its gradient maps on as they are instead of multiplying 2 x N C H W elements by 1.0
UNIT_UPSTREAM = False


def _grad_buffers(pl, pc, grad_out):
    """the two gradient maps the fused pass writes: fresh tensors, or -- when the caller owns the model's plan -- the plan's
    own upstream-gradient staging buffers (pmf_net._PlanFunction.backward then has nothing to copy)"""
    if grad_out is not None and all(g is not None and g.shape == p.shape and g.dtype == p.dtype and g.device == p.device
                                    and g.is_contiguous() for g, p in zip(grad_out, (pl, pc))):
        return grad_out[0], grad_out[1]
    return torch.empty_like(pl), torch.empty_like(pc)


def _use_torch_sort():
    return False       # (rounds 3-4 A/B: torch.sort + pmf_loss_lovasz; the in-library radix sort is the product path)


class _FusedPMFLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lidar_prob, camera_prob, label, alpha, lambda_, gamma_per, tau, focal_gamma, conf_l, conf_c, grad_out=None):
        lib = L.lib()
        if not (lidar_prob.is_cuda and camera_prob.is_cuda and label.is_cuda):
            raise RuntimeError("pmf_amd fused loss: tensors must live on the GPU (no CPU fallback)")
        pl, pc = lidar_prob.detach().contiguous().float(), camera_prob.detach().contiguous().float()
        lab = label.contiguous().long()
        n, c, h, w = pl.shape
        hw, p = h * w, n * h * w
        dev = pl.device
        gl, gc = _grad_buffers(pl, pc, grad_out)
        key = torch.empty((2 * c, p), dtype=torch.float32, device=dev)
        rows = torch.empty((lib.pmf_loss_rows(p), 4), dtype=torch.float64, device=dev)
        cnt = torch.empty(c, dtype=torch.int64, device=dev)
        nb = lib.pmf_loss_chunks(p)
        bsum = torch.empty((2 * c, nb), dtype=torch.float32, device=dev)
        dots = torch.empty((2 * c, nb), dtype=torch.float64, device=dev)
        out6 = torch.empty(8, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        a = alpha.to(dev, torch.float32).contiguous()
        L.check(lib.pmf_loss_pixel(pl.data_ptr(), pc.data_ptr(), lab.data_ptr(), a.data_ptr(), n, c, hw,
                                   float(focal_gamma), float(tau), float(gamma_per), cnt.data_ptr(), gl.data_ptr(),
                                   gc.data_ptr(), key.data_ptr(), rows.data_ptr(),
                                   conf_l.data_ptr() if conf_l is not None else None,
                                   conf_c.data_ptr() if conf_c is not None else None, st), "pmf_loss_pixel")
        if _use_torch_sort():
            vals, perm = torch.sort(key, dim=1, descending=True)
            L.check(lib.pmf_loss_lovasz(perm.data_ptr(), vals.data_ptr(), lab.data_ptr(), n, c, hw, cnt.data_ptr(),
                                        float(lambda_), float(gamma_per), bsum.data_ptr(), dots.data_ptr(), rows.data_ptr(),
                                        gl.data_ptr(), gc.data_ptr(), out6.data_ptr(), st), "pmf_loss_lovasz")
        else:       # labelled pixels of the classes present only, sorted in the library (no torch.sort, no int64 indices)
            ws = _sort_workspace(lib, c, p, dev)
            L.check(lib.pmf_loss_lovasz_sort(key.data_ptr(), lab.data_ptr(), n, c, hw, cnt.data_ptr(), float(lambda_),
                                             float(gamma_per), ws.data_ptr(), bsum.data_ptr(), dots.data_ptr(),
                                             rows.data_ptr(), gl.data_ptr(), gc.data_ptr(), out6.data_ptr(), st),
                    "pmf_loss_lovasz_sort")
        ctx.save_for_backward(gl, gc)
        ctx.mark_non_differentiable(out6)
        return out6[0].clone(), out6

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        gl, gc = ctx.saved_tensors
        if UNIT_UPSTREAM:        # (the engine's own `total.backward()`: d total / d total = 1, nothing to multiply)
            return gl, gc, None, None, None, None, None, None, None, None, None
        return gl * g_total, gc * g_total, None, None, None, None, None, None, None, None, None


class _FusedWeightedLoss(torch.autograd.Function):
    """total = sum_i w_i * term_i over (foc, lov, foc_cam, lov_cam, per_p, per_q) with the weights read on the device;
    gradients w.r.t. both probability maps (analytic, HIP) and w.r.t. the weights (= the term values)."""

    @staticmethod
    def forward(ctx, lidar_prob, camera_prob, label, alpha, w6, tau, focal_gamma, conf_l, conf_c, grad_out=None):
        lib = L.lib()
        if not (lidar_prob.is_cuda and camera_prob.is_cuda and label.is_cuda and w6.is_cuda):
            raise RuntimeError("pmf_amd fused loss: tensors must live on the GPU (no CPU fallback)")
        pl, pc = lidar_prob.detach().contiguous().float(), camera_prob.detach().contiguous().float()
        lab = label.contiguous().long()
        w = w6.detach().contiguous().float()
        n, c, h, wd = pl.shape
        hw, p = h * wd, n * h * wd
        dev = pl.device
        gl, gc = _grad_buffers(pl, pc, grad_out)
        key = torch.empty((2 * c, p), dtype=torch.float32, device=dev)
        rows = torch.empty((lib.pmf_loss_rows(p), 4), dtype=torch.float64, device=dev)
        cnt = torch.empty(c, dtype=torch.int64, device=dev)
        nb = lib.pmf_loss_chunks(p)
        bsum = torch.empty((2 * c, nb), dtype=torch.float32, device=dev)
        dots = torch.empty((2 * c, nb), dtype=torch.float64, device=dev)
        out8 = torch.empty(8, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        a = alpha.to(dev, torch.float32).contiguous()
        L.check(lib.pmf_loss_pixel_w(pl.data_ptr(), pc.data_ptr(), lab.data_ptr(), a.data_ptr(), n, c, hw,
                                     float(focal_gamma), float(tau), w.data_ptr(), cnt.data_ptr(), gl.data_ptr(),
                                     gc.data_ptr(), key.data_ptr(), rows.data_ptr(),
                                     conf_l.data_ptr() if conf_l is not None else None,
                                     conf_c.data_ptr() if conf_c is not None else None, st), "pmf_loss_pixel_w")
        if _use_torch_sort():
            vals, perm = torch.sort(key, dim=1, descending=True)
            L.check(lib.pmf_loss_lovasz_w(perm.data_ptr(), vals.data_ptr(), lab.data_ptr(), n, c, hw, cnt.data_ptr(),
                                          w.data_ptr(), bsum.data_ptr(), dots.data_ptr(), rows.data_ptr(), gl.data_ptr(),
                                          gc.data_ptr(), out8.data_ptr(), st), "pmf_loss_lovasz_w")
        else:
            ws = _sort_workspace(lib, c, p, dev)
            L.check(lib.pmf_loss_lovasz_sort_w(key.data_ptr(), lab.data_ptr(), n, c, hw, cnt.data_ptr(), w.data_ptr(),
                                               ws.data_ptr(), bsum.data_ptr(), dots.data_ptr(), rows.data_ptr(),
                                               gl.data_ptr(), gc.data_ptr(), out8.data_ptr(), st), "pmf_loss_lovasz_sort_w")
        terms = torch.stack([out8[1], out8[2], out8[3], out8[4], out8[6], out8[7]])
        ctx.save_for_backward(gl, gc, terms)
        ctx.mark_non_differentiable(out8)
        return out8[0].clone(), out8

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        gl, gc, terms = ctx.saved_tensors
        if UNIT_UPSTREAM:
            return gl, gc, None, None, terms, None, None, None, None, None
        return gl * g_total, gc * g_total, None, None, terms * g_total, None, None, None, None, None


def weighted_loss_fused(lidar_prob, camera_prob, label, alpha, w6, tau=0.7, focal_gamma=2.0, conf_lidar=None,
                        conf_camera=None, grad_out=None):
    """total = w6 . (foc, lov, foc_cam, lov_cam, per_p, per_q); w6: device tensor [6] (may require grad).
    returns (total, dict of the six terms)."""
    total, o = _FusedWeightedLoss.apply(lidar_prob, camera_prob, label, alpha, w6, tau, focal_gamma, conf_lidar,
                                        conf_camera, grad_out)
    return total, {"foc": o[1], "lov": o[2], "foc_cam": o[3], "lov_cam": o[4], "per": o[6], "per_img": o[7]}


def pmf_total_loss_fused(lidar_prob, camera_prob, label, alpha, lambda_=1.0, gamma_=0.5, tau=0.7, focal_gamma=2.0,
                         conf_lidar=None, conf_camera=None, grad_out=None):
    """returns (total, {"foc","lov","foc_cam","lov_cam","per"}); conf_* (int64 [C,C], rows = prediction) are updated
    in place when given (same counts as IOUEval.addBatch(argmax, label))."""
    total, out6 = _FusedPMFLoss.apply(lidar_prob, camera_prob, label, alpha, lambda_, gamma_, tau, focal_gamma,
                                      conf_lidar, conf_camera, grad_out)
    t = {"foc": out6[1], "lov": out6[2], "foc_cam": out6[3], "lov_cam": out6[4], "per": out6[5]}
    return total, t
