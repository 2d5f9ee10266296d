"""Range (spherical) projection on MI355X -- pc_processor/dataset/preprocess/projection.py:4-86.

Same constructor, ``doProjection`` and ``cached_data`` contract as the reference; the per-point angles / pixel indices
and the "nearest point wins" scatter (the reference: argsort by decreasing depth, last writer wins) are the HIP kernels
pmf_range_project_index / pmf_range_project_gather.  Arrays come back as tensors on the projection's device (int32
indices where the reference has int32).  There is no CPU path: without the HIP library / a GPU the call raises."""
import ctypes as C

import numpy as np
import torch

from ... import _lib as L


def _f32(v):
    return float(np.float32(v))


class RangeProjection(object):
    def __init__(self, fov_up, fov_down, proj_w, proj_h, fov_left=-180, fov_right=180, device="cuda"):
        assert fov_up >= 0 and fov_down <= 0, \
            "require fov_up >= 0 and fov_down <= 0, while fov_up/fov_down is {}/{}".format(fov_up, fov_down)
        assert fov_right >= 0 and fov_left <= 0, \
            "require fov_right >= 0 and fov_left <= 0, while fov_right/fov_left is {}/{}".format(fov_right, fov_left)
        self.fov_up = fov_up / 180.0 * np.pi
        self.fov_down = fov_down / 180.0 * np.pi
        self.fov_v = abs(self.fov_up) + abs(self.fov_down)
        self.fov_left = fov_left / 180.0 * np.pi
        self.fov_right = fov_right / 180.0 * np.pi
        self.fov_h = abs(self.fov_left) + abs(self.fov_right)
        self.proj_w, self.proj_h = proj_w, proj_h
        self.device = torch.device(device)
        self.cached_data = {}

    def _index(self, pts, want_uproj=True):
        """pts: float32 [P, C] on the device -> keys u64[H*W] (+ cached uproj arrays)"""
        lib = L.lib()
        P, Cc = pts.shape
        keys = torch.empty(self.proj_h * self.proj_w, dtype=torch.int64, device=self.device)
        n = max(P, 1)
        ux = torch.empty(n, dtype=torch.int32, device=self.device)
        uy = torch.empty(n, dtype=torch.int32, device=self.device)
        ud = torch.empty(n, dtype=torch.float32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(lib.pmf_range_project_index(pts.data_ptr(), P, Cc, _f32(abs(self.fov_left)), _f32(self.fov_h),
                                            _f32(abs(self.fov_down)), _f32(self.fov_v), self.proj_h, self.proj_w,
                                            keys.data_ptr(), ux.data_ptr(), uy.data_ptr(), ud.data_ptr(), st),
                "pmf_range_project_index")
        self.cached_data = {"uproj_x_idx": ux[:P], "uproj_y_idx": uy[:P], "uproj_depth": ud[:P]}
        return keys

    def to_device(self, pointcloud):
        if isinstance(pointcloud, torch.Tensor):
            pts = pointcloud.to(self.device, torch.float32).contiguous()
        else:
            pts = torch.as_tensor(np.ascontiguousarray(pointcloud, np.float32)).to(self.device)
        if pts.dim() != 2 or pts.shape[1] < 3:
            raise ValueError("RangeProjection: expected a [P, >=3] point array, got %s" % (tuple(pts.shape),))
        if self.device.type != "cuda":
            raise RuntimeError("RangeProjection runs on the GPU only (device=%s)" % self.device)
        return pts

    def doProjection(self, pointcloud):
        """-> (proj_pointcloud f32[H,W,C], proj_range f32[H,W], proj_idx i32[H,W], proj_mask i32[H,W])"""
        pts = self.to_device(pointcloud)
        keys = self._index(pts)
        H, W, (P, Cc) = self.proj_h, self.proj_w, pts.shape
        pc = torch.empty((H, W, Cc), dtype=torch.float32, device=self.device)
        rng = torch.empty((H, W), dtype=torch.float32, device=self.device)
        idx = torch.empty((H, W), dtype=torch.int32, device=self.device)
        mask = torch.empty((H, W), dtype=torch.int32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib().pmf_range_project_gather(pts.data_ptr(), P, Cc, keys.data_ptr(), H, W, None, None, None, None,
                                                 None, mask.data_ptr(), rng.data_ptr(), idx.data_ptr(), pc.data_ptr(),
                                                 st), "pmf_range_project_gather")
        return pc, rng, idx, mask

    def loader_item(self, pointcloud, mapped_label, mean, stds):
        """fused loader path (salsanext_loader.py:53-75): -> (feature f32[5,H,W], label f32[H,W], mask i32[H,W],
        range f32[H,W]); mapped_label: int32 [P] (already through the dataset's label map) on the device."""
        pts = self.to_device(pointcloud)
        if pts.shape[1] < 4:
            raise ValueError("SalsaNext features need x, y, z, intensity")
        keys = self._index(pts)
        H, W, (P, Cc) = self.proj_h, self.proj_w, pts.shape
        feat = torch.empty((5, H, W), dtype=torch.float32, device=self.device)
        label = torch.empty((H, W), dtype=torch.float32, device=self.device)
        mask = torch.empty((H, W), dtype=torch.int32, device=self.device)
        rng = torch.empty((H, W), dtype=torch.float32, device=self.device)
        lab = mapped_label.to(self.device, torch.int32).contiguous()
        if lab.numel() != P:
            raise ValueError("label count %d != point count %d" % (lab.numel(), P))
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib().pmf_range_project_gather(pts.data_ptr(), P, Cc, keys.data_ptr(), H, W, lab.data_ptr(),
                                                 mean.data_ptr(), stds.data_ptr(), feat.data_ptr(), label.data_ptr(),
                                                 mask.data_ptr(), rng.data_ptr(), None, None, st),
                "pmf_range_project_gather")
        return feat, label, mask, rng
