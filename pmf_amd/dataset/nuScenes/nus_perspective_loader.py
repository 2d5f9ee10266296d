"""Per-camera-view frames of a nuScenes sweep on MI355X -- tasks/pmf_eval_nuscenes/nus_perspective_loader.py:5-79.

Same constructor and item layout as the reference's ``NusPerspectiveViewLoader``:

    feature f32[8,h,w] (depth, x, y, z, intensity, r, g, b), mask f32[h,w], label f32[h,w], x_data i32[K] (rows), y_data
    i32[K] (columns), depth f32[K], point_idx i64[K] (sweep indices of the K points this camera sees), f32[1] sweep size

The geometry stays where the reference has it: ``dataset.mapLidar2Camera`` (dataset_nuscenes.py:204-282, the devkit's chain
of rigid transforms, host side).  What the loader itself does -- Euclidean depth, the last-writer-wins scatter of the kept
points into the image plane (numpy fancy assignment in the reference, :44-58), the label dictionary, image / 255 -- is one
upload plus ``pmf_project_v2_scatter`` (csrc/project.hip).  No CPU path: without a GPU the item raises."""
import ctypes as C

import numpy as np
import torch
from torch.utils.data import Dataset

from ... import _lib as L
from ..perspective_view_loader import upload_packed


class NusPerspectiveViewLoader(Dataset):
    def __init__(self, dataset, config, data_len=-1, device="cuda"):
        self.dataset, self.config, self.data_len = dataset, config, data_len
        self.device = torch.device(device)
        self._lut = None

    def _label_lut(self):
        """dataset.labelMapping (np.vectorize over a dictionary, dataset_nuscenes.py:181-186) as a 256-entry table"""
        if self._lut is None:
            keys = sorted(self.dataset.map_name_from_general_index_to_segmentation_index.keys())
            lut = np.zeros(256, np.int32)
            lut[keys] = self.dataset.labelMapping(np.asarray(keys, np.uint8)[:, None])
            self._lut = torch.from_numpy(lut).to(self.device)
        return self._lut

    def __getitem__(self, index):
        if self.device.type != "cuda":
            raise RuntimeError("NusPerspectiveViewLoader runs on the GPU only (device=%s)" % self.device)
        pointcloud, sem_label, _ = self.dataset.loadDataByIndex(index)
        image = np.array(self.dataset.loadImage(index))
        seq_id, _ = self.dataset.parsePathInfoByIndex(index)
        mapped, keep = self.dataset.mapLidar2Camera(seq_id, pointcloud[:, :3], image.shape[1], image.shape[0])
        x_data = np.ascontiguousarray(mapped[:, 0].astype(np.int32))
        y_data = np.ascontiguousarray(mapped[:, 1].astype(np.int32))
        h, w = image.shape[:2]
        depth = np.linalg.norm(pointcloud[:, :3], 2, axis=1)[keep].astype(np.float32)
        src = np.flatnonzero(keep).astype(np.int32)
        K = int(src.shape[0])
        pts, sem, img, xd, yd, dep, sidx = upload_packed(
            [np.ascontiguousarray(pointcloud[:, :4], np.float32), np.ascontiguousarray(sem_label, np.uint8).reshape(-1).astype(np.int32),
             np.ascontiguousarray(image, np.uint8), x_data, y_data, depth, src], self.device)
        lut = self._label_lut()
        proj = torch.empty((10, h, w), dtype=torch.float32, device=self.device)
        pix = torch.empty(h * w, dtype=torch.int32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib().pmf_project_v2_scatter(pts.data_ptr(), sem.data_ptr(), sidx.data_ptr(), xd.data_ptr(), yd.data_ptr(),
                                               dep.data_ptr(), K, img.data_ptr(), h, w, lut.data_ptr(), 256, 0, 0, h, w,
                                               proj.data_ptr(), pix.data_ptr(), st), "pmf_project_v2_scatter")
        return (proj[:8], proj[8], proj[9], xd, yd, dep, sidx.long(),
                torch.tensor([float(pointcloud.shape[0])], device=self.device))

    def __len__(self):
        if 0 < self.data_len < len(self.dataset):
            return self.data_len
        return len(self.dataset)
