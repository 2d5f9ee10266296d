"""Perspective-view loader on MI355X (pc_processor/dataset/perspective_view_loader.py:8-146).

Same constructor / item contract as the reference.  The projection (float64 pinhole, parser.py:209-227), the
last-writer-wins scatter of (depth,x,y,z,intensity,label,mask) and the stacking with RGB/255 run as HIP kernels
(pmf_project_scatter); the validation CenterCrop+Pad runs as pmf_crop_pad.  ``dataset`` is the reference's
duck type; only ``proj_matrix`` / ``class_map_lut`` are needed from it beyond the four load functions:
    loadDataByIndex(i) -> (points f32[P,4], sem i32[P], inst)       loadImage(i) -> PIL / uint8[h,w,3]
    parsePathInfoByIndex(i) -> (seq, frame)                          proj_matrix[seq] -> f64[3,4]
    labelMapping LUT: class_map_lut i32[L]
The training-time tensor transforms (torchvision RandomHorizontalFlip(0.5) -> RandomRotation(15) -> RandomCrop -> Pad,
perspective_view_loader.py:63-69,138-141) are ONE HIP gather (pmf_flip_rotate_crop) behind ``FlipRotateCrop``: the draws
come from torch's global RNG in torchvision's order (seeding torch reproduces its parameters), the rotation is
nearest-neighbour about the image centre with zero fill.  torchvision itself is absent here: the restatement follows its
published tensor path and is checked against torch's own grid_sample (oracle/tensor_aug_ref.py), unpinned.  The point
augmentation (``pcd_aug``, augmentor.py) runs on the GPU before the projection.  ``img_aug`` (perspective_view_loader.py:
46-49,84-85: torchvision ColorJitter(*img_jitter) on the PIL image) is ``ColorJitter`` below: the draws follow
torchvision's get_params order on torch's RNG, the four pixel operations run as HIP kernels on the uint8 frame
(pmf_color_jitter) and reproduce Pillow's ImageEnhance / HSV arithmetic bit for bit (oracle/color_jitter_ref.py is pinned
to Pillow exhaustively).
"""
import ctypes as C
import os
import threading

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import _lib as L


def upload_packed(arrays, device):
    """several host arrays -> device tensors with ONE host -> device copy (a packed byte buffer, 16-byte aligned parts).
    A pageable upload blocks the calling thread until the GPU has executed it, and next to a running training step every
    such point costs the prefetch thread about a millisecond of queueing (measured: 32 ms per batch with five uploads per
    frame against 12 ms on an idle GPU).  Pinned staging is no way out on this platform: page-locked host memory is
    mapped uncached for the CPU (filling it ran at ~100 MB/s)."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    offs, n = [], 0
    for a in arrays:
        offs.append(n)
        n += (a.nbytes + 15) // 16 * 16
    buf = np.empty(max(n, 16), np.uint8)
    for a, o in zip(arrays, offs):
        buf[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    d = torch.from_numpy(buf).to(device)
    return [d[o:o + a.nbytes].view(torch.from_numpy(a[:0] if a.ndim else a.reshape(1)[:0]).dtype).reshape(a.shape)
            for a, o in zip(arrays, offs)]


def image_to_device(image_u8, device):
    """uint8 [h, w, 3] PIL image / numpy array / tensor -> contiguous device tensor."""
    if isinstance(image_u8, torch.Tensor):
        return image_u8.to(device, torch.uint8).contiguous()
    return torch.from_numpy(np.array(image_u8, dtype=np.uint8, order="C")).to(device)


class ColorJitter(object):
    """torchvision.transforms.ColorJitter(brightness, contrast, saturation, hue) for the uint8 camera frame on the device.
    Parameter ranges and draws as torchvision 0.14.1 (transforms.py ColorJitter._check_input / get_params): a number v
    means [max(0, 1 - v), 1 + v] (hue: [-v, v], |v| <= 0.5), a zero-width range draws nothing; per call
    torch.randperm(4), then one torch.empty(1).uniform_ per active range in the order brightness, contrast, saturation,
    hue.  ``draw`` / ``apply`` are split for tests."""

    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        self.ranges = (self._range(brightness, "brightness"), self._range(contrast, "contrast"),
                       self._range(saturation, "saturation"),
                       self._range(hue, "hue", center=0.0, bound=(-0.5, 0.5), clip_first_on_zero=False))

    @staticmethod
    def _range(value, name, center=1.0, bound=(0.0, float("inf")), clip_first_on_zero=True):
        if isinstance(value, (int, float)):
            if value < 0:
                raise ValueError("If {} is a single number, it must be non negative.".format(name))
            value = [center - float(value), center + float(value)]
            if clip_first_on_zero:
                value[0] = max(value[0], 0.0)
        elif isinstance(value, (tuple, list)) and len(value) == 2:
            value = [float(value[0]), float(value[1])]
        else:
            raise TypeError("{} should be a single number or a list/tuple with length 2.".format(name))
        if not bound[0] <= value[0] <= value[1] <= bound[1]:
            raise ValueError("{} values should be between {}".format(name, bound))
        return None if value[0] == value[1] == center else tuple(value)

    def draw(self):
        order = torch.randperm(4).tolist()
        factors = [None if r is None else float(torch.empty(1).uniform_(r[0], r[1])) for r in self.ranges]
        return order, factors

    def apply(self, img, order, factors):
        """img: uint8 [h, w, 3] DEVICE tensor, jittered in place (and returned)."""
        if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous()):
            raise RuntimeError("ColorJitter runs on a contiguous uint8 GPU tensor only (no CPU fallback)")
        if img.data_ptr() % 4:
            # the kernel reads dwords (include/pmf_amd.h): a frame that starts off a 4-byte boundary -- a slice batch[b] of a
            # [B, h, w, 3] tensor with h * w % 4 != 0, e.g. 375 x 1242 KITTI frames -- goes through an aligned device copy
            tmp = img.clone()
            self.apply(tmp, order, factors)
            return img.copy_(tmp)
        scratch = torch.empty(1, dtype=torch.int64, device=img.device)
        o = (C.c_int32 * 4)(*order)
        f = (C.c_double * 4)(*[0.0 if x is None else x for x in factors])
        e = (C.c_int32 * 4)(*[int(x is not None) for x in factors])
        L.check(L.lib().pmf_color_jitter(img.data_ptr(), img.shape[0], img.shape[1], o, f, e, scratch.data_ptr(),
                                         C.c_void_p(torch.cuda.current_stream(img.device).cuda_stream)),
                "pmf_color_jitter")
        return img

    def __call__(self, img):
        return self.apply(img, *self.draw())


_PROJ_TLS = threading.local()      # per thread (prefetch workers run on streams of their own): persistent projection workspaces
_CONST_CACHE = {}                  # small host constants (projection matrix, label LUT) already on the device, by content


def _proj_workspace(dev, h, w, nblk):
    """(pix_tag u32[h*w], slots u64[>= nblk + 1], generation) of pmf_project_scatter2 for this thread AND stream: allocated once per
    image size, zeroed once and whenever the generation counter wraps (4095 frames).  Keyed by the current stream as well:
    two calls of one thread on different streams may overlap on the GPU, and call N + 1's atomicMax with generation g + 1
    would overwrite the tags call N's gather pass still reads; calls on ONE stream are ordered."""
    tab = _PROJ_TLS.__dict__.setdefault("ws", {})
    key = (str(dev), h, w, int(torch.cuda.current_stream(dev).cuda_stream))
    ent = tab.get(key)
    if ent is None or ent[1].numel() < nblk + 1:
        if len(tab) > 64:        # streams come and go (prefetch workers): do not grow without bound
            tab.clear()
        ent = tab[key] = [torch.zeros(h * w, dtype=torch.int32, device=dev),
                          torch.zeros(max(nblk + 1, 256), dtype=torch.int64, device=dev), 0]
    ent[2] += 1
    if ent[2] > 4095:
        ent[0].zero_()
        ent[1].zero_()
        ent[2] = 1
    return ent


def _device_const(a, dtype, dev):
    """a small host array as a device tensor, uploaded once per distinct content"""
    a = np.ascontiguousarray(a, dtype)
    key = (str(dev), a.dtype.str, a.tobytes())
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 256:
            _CONST_CACHE.clear()
        t = _CONST_CACHE[key] = torch.from_numpy(a.reshape(-1).copy()).to(dev)
    return t


def project_frame_gpu(points, sem_label, image_u8, proj_matrix, label_lut, device="cuda", need_uproj=True):
    """-> (proj f32[10,h,w], x_data i32[K], y_data i32[K], depth f32[P], keep bool[P]) on `device`.
    need_uproj=False (training / validation items, which return only the image-plane tensors): x_data / y_data come back
    un-trimmed (length P, the first K entries valid) so that no device -> host read of K stalls the stream."""
    lib = L.lib()
    dev = torch.device(device)
    if isinstance(points, torch.Tensor):
        pts = points.to(dev, torch.float32).contiguous()
    else:
        pts = torch.as_tensor(np.ascontiguousarray(points, np.float32)).to(dev)
    sem = sem_label.to(dev, torch.int32).contiguous() if isinstance(sem_label, torch.Tensor) else \
        torch.as_tensor(np.ascontiguousarray(sem_label, np.int32)).to(dev)
    img = image_to_device(image_u8, dev)
    # calibration matrix / label LUT: per-sequence constants -- callers that loop over frames pass device tensors
    # calibration matrix / label LUT: per-sequence constants -- device tensors are taken as they are, host arrays are
    # uploaded once per distinct content (_device_const)
    mat = proj_matrix.to(dev, torch.float64).reshape(12).contiguous() if isinstance(proj_matrix, torch.Tensor) else \
        _device_const(np.asarray(proj_matrix, np.float64).reshape(12), np.float64, dev)
    lut = label_lut.to(dev, torch.int32).contiguous() if isinstance(label_lut, torch.Tensor) else \
        _device_const(label_lut, np.int32, dev)
    P = pts.shape[0]
    h, w = img.shape[0], img.shape[1]
    out = torch.empty((10, h, w), dtype=torch.float32, device=dev)
    # keep / x / y / depth / count: ONE allocation (they go back to the caller, so they cannot live in the cached workspace)
    n1 = max(P, 1)
    blob = torch.empty(3 * n1 + (n1 + 3) // 4 + 4, dtype=torch.int32, device=dev)
    xd, yd = blob[:n1], blob[n1:2 * n1]
    depth = blob[2 * n1:3 * n1].view(torch.float32)
    keep = blob[3 * n1:3 * n1 + (n1 + 3) // 4].view(torch.uint8)[:n1]
    nk = blob[3 * n1 + (n1 + 3) // 4:3 * n1 + (n1 + 3) // 4 + 1]
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = L.PMF_E_UNSUPPORTED if os.environ.get("PMF_PROJECT_LEGACY") == "1" else 0
    if rc == 0:
        # two launches, no memset, persistent per-thread workspace (csrc/project.hip pmf_project_scatter2)
        nblk = (P + 1023) // 1024 + 1
        pix, slots, gen = _proj_workspace(dev, h, w, nblk)
        rc = lib.pmf_project_scatter2(pts.data_ptr(), sem.data_ptr(), P, img.data_ptr(), h, w, mat.data_ptr(),
                                      lut.data_ptr(), lut.shape[0], out.data_ptr(), keep.data_ptr(), xd.data_ptr(),
                                      yd.data_ptr(), depth.data_ptr(), nk.data_ptr(), pix.data_ptr(), slots.data_ptr(),
                                      gen, C.c_void_p(stream))
    if rc == L.PMF_E_UNSUPPORTED:      # more than 2^20 points (or PMF_PROJECT_LEGACY=1): the five-launch form
        pix = torch.empty(h * w, dtype=torch.int32, device=dev)
        blk = torch.empty((P + 1023) // 1024 + 1, dtype=torch.int32, device=dev)
        rc = lib.pmf_project_scatter(pts.data_ptr(), sem.data_ptr(), P, img.data_ptr(), h, w, mat.data_ptr(),
                                     lut.data_ptr(), lut.shape[0], out.data_ptr(), keep.data_ptr(), xd.data_ptr(),
                                     yd.data_ptr(), depth.data_ptr(), nk.data_ptr(), pix.data_ptr(), blk.data_ptr(),
                                     C.c_void_p(stream))
    L.check(rc, "pmf_project_scatter")
    if not need_uproj:
        return out, xd, yd, depth[:P], keep[:P].bool()
    k = int(nk.item())
    if k < 0:       # the projection kernel found its workspace in a bad state (csrc/project.hip: ticket beyond the grid)
        _PROJ_TLS.__dict__.pop("ws", None)
        raise RuntimeError("pmf_project_scatter2: corrupt ticket workspace (it has been dropped; the next call starts clean)")
    return out, xd[:k], yd[:k], depth[:P], keep[:P].bool()


def center_crop_pad_gpu(proj, out_h, out_w, h_pad, w_pad):
    """CenterCrop((out_h-2hp, out_w-2wp)) then Pad((wp, hp)) (perspective_view_loader.py:71-74,138-141)."""
    lib = L.lib()
    c, h, w = proj.shape
    ch, cw = out_h - 2 * h_pad, out_w - 2 * w_pad
    # torchvision CenterCrop: symmetric zero-pad first when the image is smaller than the crop
    pl = (cw - w) // 2 if cw > w else 0
    pt = (ch - h) // 2 if ch > h else 0
    hh = max(h, ch) if ch > h else h
    ww = max(w, cw) if cw > w else w
    top = int(round((hh - ch) / 2.0)) - pt
    left = int(round((ww - cw) / 2.0)) - pl
    dst = torch.empty((c, out_h, out_w), dtype=torch.float32, device=proj.device)
    stream = torch.cuda.current_stream(proj.device).cuda_stream
    rc = lib.pmf_crop_pad(proj.contiguous().data_ptr(), c, h, w, top, left, dst.data_ptr(), out_h, out_w, h_pad,
                          w_pad, ch, cw, C.c_void_p(stream))
    L.check(rc, "pmf_crop_pad")
    return dst


class FlipRotateCrop(object):
    """the training ``aug_ops`` (+ Pad) of the reference as one call: [C,h,w] device tensor -> [C, crop_h + 2*h_pad,
    crop_w + 2*w_pad].  ``draw`` exposes the parameters (flip, angle, top, left) for tests."""

    def __init__(self, crop_h, crop_w, h_pad=0, w_pad=0, p=0.5, degrees=15.0):
        self.crop_h, self.crop_w, self.h_pad, self.w_pad, self.p, self.degrees = crop_h, crop_w, h_pad, w_pad, p, degrees

    def draw(self, h, w):
        flip = bool(torch.rand(1) < self.p)
        angle = float(torch.empty(1).uniform_(-float(self.degrees), float(self.degrees)).item())
        if h < self.crop_h or w < self.crop_w:
            raise ValueError("Required crop size {} is larger than input image size {}".format(
                (self.crop_h, self.crop_w), (h, w)))
        if (h, w) == (self.crop_h, self.crop_w):
            return flip, angle, 0, 0
        top = int(torch.randint(0, h - self.crop_h + 1, size=(1,)).item())
        left = int(torch.randint(0, w - self.crop_w + 1, size=(1,)).item())
        return flip, angle, top, left

    def apply(self, proj, flip, angle, top, left):
        import math
        if not proj.is_cuda:
            raise RuntimeError("FlipRotateCrop runs on the GPU only (no CPU fallback)")
        c, h, w = proj.shape
        r = math.radians(-angle)
        m = (C.c_float * 6)(math.cos(r), math.sin(r), 0.0, -math.sin(r), math.cos(r), 0.0)
        oh, ow = self.crop_h + 2 * self.h_pad, self.crop_w + 2 * self.w_pad
        dst = torch.empty((c, oh, ow), dtype=torch.float32, device=proj.device)
        src = proj.contiguous().float()
        L.check(L.lib().pmf_flip_rotate_crop(src.data_ptr(), c, h, w, int(flip), m, top, left, self.crop_h, self.crop_w,
                                             self.h_pad, self.w_pad, dst.data_ptr(), oh, ow,
                                             C.c_void_p(torch.cuda.current_stream(proj.device).cuda_stream)),
                "pmf_flip_rotate_crop")
        return dst

    def __call__(self, proj):
        return self.apply(proj, *self.draw(proj.shape[1], proj.shape[2]))


class PerspectiveViewLoader(Dataset):
    def __init__(self, dataset, config, data_len=-1, is_train=True, pcd_aug=False, img_aug=False,
                 use_padding=False, return_uproj=False, device="cuda", aug_ops=None):
        self.dataset, self.config = dataset, config
        self.is_train, self.data_len = is_train, data_len
        self.use_padding, self.return_uproj = use_padding, return_uproj
        self.device = device
        self.aug_ops = aug_ops
        self.img_aug = bool(img_aug and is_train)                  # perspective_view_loader.py:19-21,46-49
        self.img_jitter = ColorJitter(*config["augmentation"]["img_jitter"]) if self.img_aug else None
        self.pcd_aug = bool(pcd_aug and is_train)
        self.augmentor = None
        if self.pcd_aug:                    # perspective_view_loader.py:24-41
            from .preprocess import augmentor
            a = config["augmentation"]
            params = augmentor.AugmentParams()
            params.setFlipProb(p_flipx=a["p_flipx"], p_flipy=a["p_flipy"])
            params.setTranslationParams(**{k: a[k] for k in a if "trans" in k})
            params.setRotationParams(**{k: a[k] for k in a if "rot" in k})
            self.augmentor = augmentor.Augmentor(params, device=device)
        s = config["sensor"]
        self.h_pad = s["h_pad"] if use_padding else 0
        self.w_pad = s["w_pad"] if use_padding else 0
        self.out_h = s["proj_ht"] if is_train else s["proj_h"]
        self.out_w = s["proj_wt"] if is_train else s["proj_w"]
        self._own_aug = False
        # __getitem__ may run on several prefetch threads at once (tasks/pmf/trainer.py Prefetcher): what it leaves behind for
        # the caller (last_keep) is per thread, and the device-constant cache is filled under a lock
        import threading
        self._tls, self._const_lock = threading.local(), threading.Lock()
        if is_train and aug_ops is None:
            # RandomCrop(size=(proj_ht - 2*h_pad, proj_wt - 2*w_pad)) then Pad((w_pad, h_pad)), both inside the gather
            self.aug_ops = FlipRotateCrop(s["proj_ht"] - 2 * self.h_pad, s["proj_wt"] - 2 * self.w_pad, self.h_pad, self.w_pad)
            self._own_aug = True

    @property
    def last_keep(self):
        """bool[P] of the frame THIS thread projected last: the points behind x_data / y_data, in file order"""
        return getattr(self._tls, "keep", None)

    @last_keep.setter
    def last_keep(self, keep):
        self._tls.keep = keep

    def __getitem__(self, index):
        pointcloud, sem_label, _ = self.dataset.loadDataByIndex(index)
        image = self.dataset.loadImage(index)
        if not isinstance(pointcloud, torch.Tensor) and not isinstance(image, torch.Tensor):
            # the frame's three host arrays in one upload (see upload_packed)
            pointcloud, sem_label, image = upload_packed(
                [np.asarray(pointcloud, np.float32), np.asarray(sem_label, np.int32), np.array(image, dtype=np.uint8, order="C")],
                self.device)
        if self.pcd_aug:
            pointcloud = self.augmentor.doAugmentation(pointcloud)
        image = image_to_device(image, self.device)
        if self.img_aug:
            image = self.img_jitter(image)
        seq_id, _ = self.dataset.parsePathInfoByIndex(index)
        proj, xd, yd, depth, keep = project_frame_gpu(pointcloud, sem_label, image, self._const("mat", seq_id),
                                                      self._const("lut", None), self.device,
                                                      need_uproj=self.return_uproj)
        self.last_keep = keep            # bool[P]: the points behind x_data / y_data, in file order
        if self.return_uproj:
            return proj[:8], proj[8], proj[9], xd, yd, depth
        if self.is_train:
            proj = self.aug_ops(proj)
            if self.use_padding and not self._own_aug:
                proj = torch.nn.functional.pad(proj, (self.w_pad, self.w_pad, self.h_pad, self.h_pad))
        else:
            proj = center_crop_pad_gpu(proj, self.out_h, self.out_w, self.h_pad, self.w_pad)
        return proj[:8], proj[8], proj[9]

    def _const(self, kind, seq_id):
        """the sequence's projection matrix / the label LUT as device tensors, uploaded once"""
        key = (kind, seq_id)
        with self._const_lock:
            cache = self.__dict__.setdefault("_dev_const", {})
            if key not in cache:
                if kind == "mat":
                    cache[key] = torch.as_tensor(np.ascontiguousarray(self.dataset.proj_matrix[seq_id], np.float64).reshape(12)).to(self.device)
                else:
                    cache[key] = torch.as_tensor(np.ascontiguousarray(self.dataset.class_map_lut, np.int32)).to(self.device)
            return cache[key]

    def __len__(self):
        if 0 < self.data_len < len(self.dataset):
            return self.data_len
        return len(self.dataset)
