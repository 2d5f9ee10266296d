"""Sweep augmentation -- pc_processor/dataset/preprocess/augmentor.py:6-180.

The random draws stay on the host in the reference's order (Python's ``random``; seeding it reproduces the reference's
sequence), the Euler matrix comes from scipy as in the reference, and the rigid transform itself -- flips, float32
translation, float64 rotation rounded to float32 -- is one HIP kernel over the sweep (pmf_points_transform), so the
points never leave the GPU between augmentation and projection."""
import ctypes as C
import random

import numpy as np
import torch

from ... import _lib as L

_FIELDS = ("p_flipx", "p_flipy", "p_transx", "trans_xmin", "trans_xmax", "p_transy", "trans_ymin", "trans_ymax",
           "p_transz", "trans_zmin", "trans_zmax", "p_rot_roll", "rot_rollmin", "rot_rollmax", "p_rot_pitch",
           "rot_pitchmin", "rot_pitchmax", "p_rot_yaw", "rot_yawmin", "rot_yawmax")


class AugmentParams(object):
    def __init__(self, **kw):
        for f in _FIELDS:
            setattr(self, f, kw.pop(f, 0.))
        if kw:
            raise TypeError("unknown augmentation parameter(s): %s" % sorted(kw))

    def setFlipProb(self, p_flipx, p_flipy):
        self.p_flipx, self.p_flipy = p_flipx, p_flipy

    def _set(self, allowed, kw):
        for k, v in kw.items():
            if k not in allowed:
                raise TypeError("unknown parameter %r" % k)
            setattr(self, k, v)

    def setTranslationParams(self, **kw):
        self._set([f for f in _FIELDS if "trans" in f], kw)

    def setRotationParams(self, **kw):
        self._set([f for f in _FIELDS if "rot" in f], kw)

    def __str__(self):
        return "=== Augmentor parameters ===\n" + "\n".join("%s: %s" % (f, getattr(self, f)) for f in _FIELDS)


def _euler_zyx(yaw, pitch, roll, degrees=True):
    from scipy.spatial.transform import Rotation
    return Rotation.from_euler("zyx", [yaw, pitch, roll], degrees=degrees).as_matrix()


class Augmentor(object):
    def __init__(self, params, device="cuda"):
        self.parmas = params           # (sic) the reference's attribute name
        self.device = torch.device(device)

    def _to_device(self, pointcloud):
        if isinstance(pointcloud, torch.Tensor):
            return pointcloud.to(self.device, torch.float32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(pointcloud, np.float32)).to(self.device)

    def transform(self, pointcloud, flipx=False, flipy=False, trans=(0, 0, 0), rot_matrix=None):
        """in place on a device tensor (a numpy input is copied to the device first) -> device tensor"""
        pts = self._to_device(pointcloud)
        if self.device.type != "cuda":
            raise RuntimeError("Augmentor runs on the GPU only (device=%s)" % self.device)
        rot = None
        if rot_matrix is not None:
            rot = torch.as_tensor(np.ascontiguousarray(rot_matrix, np.float64).reshape(9)).to(self.device)
        t = [float(np.float32(v)) for v in trans]
        L.check(L.lib().pmf_points_transform(pts.data_ptr(), pts.shape[0], pts.shape[1], int(bool(flipx)),
                                             int(bool(flipy)), t[0], t[1], t[2], rot.data_ptr() if rot is not None else None,
                                             C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "pmf_points_transform")
        return pts

    def flipX(self, pointcloud):
        return self.transform(pointcloud, flipx=True)

    def flipY(self, pointcloud):
        return self.transform(pointcloud, flipy=True)

    def translation(self, pointcloud, x, y, z):
        return self.transform(pointcloud, trans=(x, y, z))

    def rotation(self, pointcloud, roll, pitch, yaw, degrees=True):
        return self.transform(pointcloud, rot_matrix=_euler_zyx(yaw, pitch, roll, degrees))

    def draw(self):
        """the reference's sequence of random.uniform calls (augmentor.py:123-178)"""
        p = self.parmas

        def maybe(prob, lo, hi):
            return random.uniform(lo, hi) if random.uniform(0, 1) < prob else 0
        flipx = random.uniform(0, 1) < p.p_flipx
        flipy = random.uniform(0, 1) < p.p_flipy
        tx = maybe(p.p_transx, p.trans_xmin, p.trans_xmax)
        ty = maybe(p.p_transy, p.trans_ymin, p.trans_ymax)
        tz = maybe(p.p_transz, p.trans_zmin, p.trans_zmax)
        roll = maybe(p.p_rot_roll, p.rot_rollmin, p.rot_rollmax)
        pitch = maybe(p.p_rot_pitch, p.rot_pitchmin, p.rot_pitchmax)
        yaw = maybe(p.p_rot_yaw, p.rot_yawmin, p.rot_yawmax)
        return flipx, flipy, (tx, ty, tz), (roll, pitch, yaw)

    def doAugmentation(self, pointcloud):
        flipx, flipy, trans, (roll, pitch, yaw) = self.draw()
        return self.transform(pointcloud, flipx, flipy, trans, _euler_zyx(yaw, pitch, roll))
