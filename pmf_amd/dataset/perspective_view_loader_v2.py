"""EPMF perspective-view loader on MI355X (pc_processor/dataset/perspective_view_loader_v2.py:9-163).

Same constructor / item contract as the reference for the deterministic paths.  The yaw-cropped float64 projection
(parser.py:229-257), the order-preserving compaction, the bounding box, the last-writer-wins scatter and the RGB window
are HIP kernels (pmf_project_v2_index / pmf_project_v2_scatter); the frame size is data dependent, so one scalar read
(kept count + bounding box) separates the two passes, exactly where the reference computes min / max on the host.
``dataset`` duck type: loadDataByIndex, loadImage, parsePathInfoByIndex, proj_matrix[seq], class_map_lut and
optionally fov_left / fov_right (defaults: +-45 degrees, parser.py:36-37).
Training path (:25-34,50-57,142-153): random image rescale in [1, 1.2] (numpy RNG draw, PIL bilinear resize on the host
as in the reference), coordinates scaled with it inside the projection kernel, bottom / centred horizontal zero padding
to (proj_ht, proj_wt), then flip / rotate(15) / crop as one HIP gather (FlipRotateCrop).  img_aug (:19-23,46-47,
ColorJitter(*PVconfig.img_jitter) before the rescale) runs on the device (perspective_view_loader.ColorJitter ->
pmf_color_jitter, bit-exact Pillow arithmetic); the jittered uint8 frame comes back once for the host-side PIL resize."""
import ctypes as C
import math

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import _lib as L


def project_frame_v2_gpu(points, sem_label, image_u8, proj_matrix, label_lut, fov_left, fov_right, device="cuda",
                         img_scale=1.0):
    """-> (proj f32[10,h,w], xy_index f64[K,2], depth f32[K], keep bool[P]) on `device`; img_scale multiplies the
    projected (row, col) coordinates (the caller passes the image already rescaled by it)."""
    lib = L.lib()
    dev = torch.device(device)
    from .perspective_view_loader import image_to_device
    pts = points.to(dev, torch.float32).contiguous() if isinstance(points, torch.Tensor) else \
        torch.as_tensor(np.ascontiguousarray(points, np.float32)).to(dev)
    sem = sem_label.to(dev, torch.int32).contiguous() if isinstance(sem_label, torch.Tensor) else \
        torch.as_tensor(np.ascontiguousarray(sem_label, np.int32)).to(dev)
    img = image_to_device(image_u8, dev)
    mat = torch.as_tensor(np.ascontiguousarray(proj_matrix, np.float64).reshape(12)).to(dev)
    lut = torch.as_tensor(np.ascontiguousarray(label_lut, np.int32)).to(dev)
    P = pts.shape[0]
    n = max(P, 1)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    src = torch.empty(n, dtype=torch.int32, device=dev)
    xd = torch.empty(n, dtype=torch.int32, device=dev)
    yd = torch.empty(n, dtype=torch.int32, device=dev)
    xy = torch.empty((n, 2), dtype=torch.float64, device=dev)
    depth = torch.empty(n, dtype=torch.float32, device=dev)
    meta = torch.zeros(5, dtype=torch.int32, device=dev)          # n_kept, row_min, row_max, col_min, col_max
    blk = torch.empty((P + 1023) // 1024 + 1, dtype=torch.int32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(lib.pmf_project_v2_index_scaled(pts.data_ptr(), P, mat.data_ptr(), float(fov_left), float(fov_right),
                                     float(img_scale), keep.data_ptr(), src.data_ptr(), xd.data_ptr(), yd.data_ptr(), xy.data_ptr(),
                                     depth.data_ptr(), meta.data_ptr(), meta.data_ptr() + 4, blk.data_ptr(), st),
            "pmf_project_v2_index_scaled")
    k, x_min, x_max, y_min, y_max = [int(v) for v in meta.tolist()]    # the one host read (frame size is data)
    if k == 0:
        raise ValueError("PerspectiveViewLoaderV2: no point inside the yaw field of view")
    h, w = x_max - x_min + 1, y_max - y_min + 1
    out = torch.empty((10, h, w), dtype=torch.float32, device=dev)
    pix = torch.empty(h * w, dtype=torch.int32, device=dev)
    L.check(lib.pmf_project_v2_scatter(pts.data_ptr(), sem.data_ptr(), src.data_ptr(), xd.data_ptr(), yd.data_ptr(),
                                       depth.data_ptr(), k, img.data_ptr(), img.shape[0], img.shape[1], lut.data_ptr(),
                                       lut.shape[0], x_min, y_min, h, w, out.data_ptr(), pix.data_ptr(), st),
            "pmf_project_v2_scatter")
    return out, xy[:k], depth[:k], keep[:P].bool()


class PerspectiveViewLoaderV2(Dataset):
    def __init__(self, dataset, config, data_len=-1, is_train=True, img_aug=False, return_uproj=False, device="cuda"):
        self.dataset, self.config = dataset, config
        self.is_train, self.img_aug, self.data_len = is_train, img_aug, data_len
        self.pv_config = config["PVconfig"]
        self.return_uproj, self.device = return_uproj, device
        self.img_jitter = None
        if img_aug:                       # :19-23 (the reference jitters whenever img_aug is set, train or not)
            from .perspective_view_loader import ColorJitter
            self.img_jitter = ColorJitter(*self.pv_config["img_jitter"])
        self.aug_ops = None
        if is_train:                      # :25-34 flip / rotate(15) / crop to (proj_ht, proj_wt) as one HIP gather
            from .perspective_view_loader import FlipRotateCrop
            self.aug_ops = FlipRotateCrop(self.pv_config["proj_ht"], self.pv_config["proj_wt"])

    def __getitem__(self, index):
        image = self.dataset.loadImage(index)
        if self.img_jitter is not None:
            from .perspective_view_loader import image_to_device
            image = self.img_jitter(image_to_device(image, self.device))
            if self.is_train:
                image = image.cpu().numpy()           # the rescale below is PIL's (host), as in the reference
        img_scale = 1.0
        if self.is_train:
            # :50-57 random rescale of the camera image: numpy's global RNG, PIL's bilinear resize (what
            # torchvision's Resize calls for a PIL image) -- host image decoding side, as in the reference
            from PIL import Image
            if not isinstance(image, Image.Image):
                image = Image.fromarray(np.asarray(image))
            img_w, img_h = image.size
            img_scale = np.random.uniform(low=1.0, high=1.2)
            image = image.resize((int(img_w * img_scale), int(img_h * img_scale)), Image.BILINEAR)
        if not isinstance(image, torch.Tensor):
            image = np.asarray(image)
        pointcloud, sem_label, _ = self.dataset.loadDataByIndex(index)
        seq_id, _ = self.dataset.parsePathInfoByIndex(index)
        fl = getattr(self.dataset, "fov_left", -45 / 180.0 * math.pi)
        fr = getattr(self.dataset, "fov_right", 45 / 180.0 * math.pi)
        if not self.return_uproj and not isinstance(image, torch.Tensor) and not isinstance(pointcloud, torch.Tensor):
            from .perspective_view_loader import upload_packed     # one host -> device copy per frame
            pointcloud, sem_label, image = upload_packed(
                [np.asarray(pointcloud, np.float32), np.asarray(sem_label, np.int32), np.ascontiguousarray(image, np.uint8)],
                self.device)
        proj, xy, depth, keep = project_frame_v2_gpu(pointcloud, sem_label, image, self.dataset.proj_matrix[seq_id],
                                                     self.dataset.class_map_lut, fl, fr, self.device, img_scale)
        if self.return_uproj:
            return proj, xy, depth, keep, torch.as_tensor(np.asarray(pointcloud))
        ch, cw = (self.pv_config["proj_ht"], self.pv_config["proj_wt"]) if self.is_train else \
            (self.pv_config["proj_h"], self.pv_config["proj_w"])
        _, h, w = proj.shape
        mh, mw = max(ch, h), max(cw, w)
        left = (mw - w) // 2
        padded = torch.nn.functional.pad(proj, (left, mw - w - left, 0, mh - h))      # :142-147
        if self.is_train:
            return self.aug_ops(padded)
        top, lft = int(round((mh - ch) / 2.0)), int(round((mw - cw) / 2.0))            # CenterCrop (:36-39)
        return padded[:, top:top + ch, lft:lft + cw].contiguous()

    def __len__(self):
        if 0 < self.data_len < len(self.dataset):
            return self.data_len
        return len(self.dataset)
