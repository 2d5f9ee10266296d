"""KNN post-processing on MI355X -- same call surface as pc_processor/postproc/knn.py:37-143.

``KNN(params, nclasses)(proj_range[H,W] f32, unproj_range[P] f32, proj_argmax[H,W] i64, px[P] i64, py[P] i64)
-> labels[P] i64``.  One HIP launch (pmf_knn_vote); the reference's two [1,S*S,H*W] unfolds and four
[1,S*S,P] gathers are never materialised.  No CPU fallback: inputs must live on the GPU.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


def inverse_gaussian_window(search, sigma):
    """1 - normalised Gaussian (knn.py:12-34,103-105), evaluated with the reference's float32 torch ops on
    the host so the S*S weights are bit-identical to the reference's."""
    c = torch.arange(search)
    xg = c.repeat(search).view(search, search)
    xy = torch.stack([xg, xg.t()], dim=-1).float()
    mean = (search - 1) / 2.
    var = sigma ** 2.
    g = (1. / (2. * math.pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2., dim=-1) / (2 * var))
    g = g / torch.sum(g)
    return (1 - g).reshape(-1).contiguous()


class KNN(nn.Module):
    def __init__(self, params, nclasses):
        super().__init__()
        self.knn = params["knn"]
        self.search = params["search"]
        self.sigma = params["sigma"]
        self.cutoff = params["cutoff"]
        self.nclasses = nclasses
        self._w = None

    def forward(self, proj_range, unproj_range, proj_argmax, px, py):
        if self.search % 2 == 0:
            raise ValueError("Nearest neighbor kernel must be odd number")     # knn.py:73-74
        if not proj_range.is_cuda:
            raise RuntimeError("pmf_amd KNN runs on the GPU only (no CPU fallback)")
        lib = L.lib()
        dev = proj_range.device
        H, W = proj_range.shape
        P = unproj_range.shape[0]
        if px.shape[0] != P or py.shape[0] != P:
            raise ValueError("len(unproj_range) must equal len(px) == len(py): %d vs %d/%d"
                             % (P, px.shape[0], py.shape[0]))
        if self._w is None or self._w.device != dev:
            self._w = inverse_gaussian_window(self.search, self.sigma).to(dev)
        pr = proj_range.contiguous().float()
        ur = unproj_range.contiguous().float()
        am = proj_argmax.contiguous().long()
        pxx, pyy = px.contiguous().long(), py.contiguous().long()
        out = torch.empty(P, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.pmf_knn_vote(pr.data_ptr(), ur.data_ptr(), am.data_ptr(), pxx.data_ptr(), pyy.data_ptr(),
                              H, W, P, int(self.knn), int(self.search), self._w.data_ptr(),
                              C.c_float(float(self.cutoff)), int(self.nclasses), out.data_ptr(),
                              C.c_void_p(stream))
        L.check(rc, "pmf_knn_vote")
        return out

    def forward_batch(self, proj_range, proj_argmax, unproj_range, px, py, offsets):
        """all frames of a batch in ONE launch (pmf_knn_vote_batch): proj_range f32[B,H,W], proj_argmax i64[B,H,W]; the
        points of all frames concatenated (unproj_range f32[P], px / py i64[P]), frame b owning [offsets[b], offsets[b+1])
        (offsets: int64[B+1], device).  Returns labels i64[P]; bit-identical to B calls of forward()."""
        if self.search % 2 == 0:
            raise ValueError("Nearest neighbor kernel must be odd number")
        if not proj_range.is_cuda:
            raise RuntimeError("pmf_amd KNN runs on the GPU only (no CPU fallback)")
        dev = proj_range.device
        B, H, W = proj_range.shape
        P = unproj_range.shape[0]
        if px.shape[0] != P or py.shape[0] != P or offsets.shape[0] != B + 1 or tuple(proj_argmax.shape) != (B, H, W):
            raise ValueError("forward_batch: inconsistent shapes")
        if self._w is None or self._w.device != dev:
            self._w = inverse_gaussian_window(self.search, self.sigma).to(dev)
        pr, am = proj_range.contiguous().float(), proj_argmax.contiguous().long()
        ur, pxx, pyy = unproj_range.contiguous().float(), px.contiguous().long(), py.contiguous().long()
        off = offsets.contiguous().long()
        out = torch.empty(P, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = L.lib().pmf_knn_vote_batch(pr.data_ptr(), ur.data_ptr(), am.data_ptr(), pxx.data_ptr(), pyy.data_ptr(),
                                        off.data_ptr(), B, H, W, P, int(self.knn), int(self.search), self._w.data_ptr(),
                                        C.c_float(float(self.cutoff)), int(self.nclasses), out.data_ptr(),
                                        C.c_void_p(stream))
        L.check(rc, "pmf_knn_vote_batch")
        return out

    def forward_batch_prob(self, proj_range, prob, unproj_range, px, py, offsets, bind=False):
        """forward_batch straight from the network's probability maps prob f32[B, nclasses, H, W] (pmf_knn_vote_batch_prob): the
        class argmax is taken inside the library as an int32 map -- no torch.argmax launch, no int64 [B, H, W] tensor
        (tasks/pmf_eval_semantickitti/infer.py:96-112).  Labels bit-identical to forward_batch(..., prob.argmax(1), ...).
        bind=True: returns (fn, labels) like bind_batch."""
        if self.search % 2 == 0:
            raise ValueError("Nearest neighbor kernel must be odd number")
        if not (proj_range.is_cuda and prob.is_cuda):
            raise RuntimeError("pmf_amd KNN runs on the GPU only (no CPU fallback)")
        dev = proj_range.device
        B, H, W = proj_range.shape
        P = unproj_range.shape[0]
        if px.shape[0] != P or py.shape[0] != P or offsets.shape[0] != B + 1 or tuple(prob.shape) != (B, self.nclasses, H, W):
            raise ValueError("forward_batch_prob: inconsistent shapes")
        if self._w is None or self._w.device != dev:
            self._w = inverse_gaussian_window(self.search, self.sigma).to(dev)
        keep = (proj_range.contiguous().float(), prob.contiguous().float(), unproj_range.contiguous().float(),
                px.contiguous().long(), py.contiguous().long(), offsets.contiguous().long(),
                torch.empty(B * H * W, dtype=torch.int32, device=dev), torch.empty(P, dtype=torch.int64, device=dev), self._w)
        fn_c = L.lib().pmf_knn_vote_batch_prob
        args = (keep[0].data_ptr(), keep[2].data_ptr(), keep[1].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                keep[5].data_ptr(), B, H, W, P, int(self.knn), int(self.search), self._w.data_ptr(),
                C.c_float(float(self.cutoff)), int(self.nclasses), keep[6].data_ptr(), keep[7].data_ptr())

        def fn(_keep=keep):
            return fn_c(*args, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        L.check(fn(), "pmf_knn_vote_batch_prob")
        return (fn, keep[7]) if bind else keep[7]

    def bind_batch(self, proj_range, proj_argmax, unproj_range, px, py, offsets):
        """the launch of forward_batch with every argument resolved once: returns (fn, labels) where fn() enqueues the
        kernel on the current stream and nothing else (no allocation, no checks) -- for loops over fixed buffers and for
        timing the kernel rather than the wrapper (the checks / allocation of forward_batch cost ~15 us of host time)."""
        out = self.forward_batch(proj_range, proj_argmax, unproj_range, px, py, offsets)
        B, H, W = proj_range.shape
        keep = (proj_range.contiguous().float(), proj_argmax.contiguous().long(), unproj_range.contiguous().float(),
                px.contiguous().long(), py.contiguous().long(), offsets.contiguous().long(), out, self._w)
        fn_c = L.lib().pmf_knn_vote_batch
        args = (keep[0].data_ptr(), keep[2].data_ptr(), keep[1].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                keep[5].data_ptr(), B, H, W, unproj_range.shape[0], int(self.knn), int(self.search), self._w.data_ptr(),
                C.c_float(float(self.cutoff)), int(self.nclasses), out.data_ptr())
        dev = proj_range.device

        def fn(_keep=keep):
            return fn_c(*args, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        return fn, out

    def batch_prob(self, prob, frames):
        """prob f32[B, nclasses, H, W]; frames: list of (proj_range[H,W], unproj_range[P_b], px[P_b], py[P_b]) -> list of label
        tensors; argmax + vote inside the library (forward_batch_prob)."""
        n = [f[1].shape[0] for f in frames]
        dev = prob.device
        off = torch.tensor([0] + list(np.cumsum(n)), dtype=torch.int64, device=dev)
        out = self.forward_batch_prob(torch.stack([f[0] for f in frames]), prob, torch.cat([f[1] for f in frames]),
                                      torch.cat([f[2] for f in frames]), torch.cat([f[3] for f in frames]), off)
        return list(out.split(n))

    def batch(self, frames):
        """frames: list of (proj_range[H,W], unproj_range[P_b], proj_argmax[H,W], px[P_b], py[P_b]) of equal H, W -> list of
        label tensors, one launch for all of them."""
        n = [f[1].shape[0] for f in frames]
        dev = frames[0][0].device
        off = torch.tensor([0] + list(np.cumsum(n)), dtype=torch.int64, device=dev)
        out = self.forward_batch(torch.stack([f[0] for f in frames]), torch.stack([f[2] for f in frames]),
                                 torch.cat([f[1] for f in frames]), torch.cat([f[3] for f in frames]),
                                 torch.cat([f[4] for f in frames]), off)
        return list(out.split(n))
