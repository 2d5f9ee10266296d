"""SalsaNext range-image loader on MI355X -- pc_processor/dataset/salsanext_loader.py:7-89.

Same constructor / item contract as the reference: (proj_feature [5,H,W], proj_sem_label [H,W], proj_mask [H,W]) and,
with return_uproj, (+ proj_range, uproj_x, uproj_y, uproj_depth).  The sweep goes to the GPU once; augmentation
(pmf_points_transform), projection (pmf_range_project_index) and the tensor assembly incl. normalisation and label
lookup (pmf_range_project_gather) are HIP kernels, and the item stays on the device -- feed it to SalsaNext directly.
``dataset`` duck type: loadDataByIndex(i) -> (pointcloud [P,4], sem_label [P], inst_label), labelMapping(labels)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from .preprocess import augmentor, projection


class SalsaNextLoader(Dataset):
    def __init__(self, dataset, config, data_len=-1, is_train=True, return_uproj=False, device="cuda"):
        self.dataset, self.config = dataset, config
        self.is_train, self.data_len, self.return_uproj = is_train, data_len, return_uproj
        self.device = torch.device(device)
        if self.is_train:
            a = self.config["augmentation"]
            params = augmentor.AugmentParams()
            params.setFlipProb(p_flipx=a["p_flipx"], p_flipy=a["p_flipy"])
            params.setTranslationParams(**{k: a[k] for k in a if "trans" in k})
            params.setRotationParams(**{k: a[k] for k in a if "rot" in k})
            self.augmentor = augmentor.Augmentor(params, device=self.device)
        else:
            self.augmentor = None
        s = self.config["sensor"]
        self.projection = projection.RangeProjection(fov_up=s["fov_up"], fov_down=s["fov_down"], fov_left=s["fov_left"],
                                                     fov_right=s["fov_right"], proj_h=s["proj_h"], proj_w=s["proj_w"],
                                                     device=self.device)
        self.proj_img_mean = torch.tensor(s["img_mean"], dtype=torch.float)
        self.proj_img_stds = torch.tensor(s["img_stds"], dtype=torch.float)
        self._mean_dev = self._stds_dev = None

    def __getitem__(self, index):
        pointcloud, sem_label, inst_label = self.dataset.loadDataByIndex(index)
        pts = self.projection.to_device(pointcloud)
        if self.is_train:
            pts = self.augmentor.doAugmentation(pts.clone() if isinstance(pointcloud, torch.Tensor) else pts)
        mapped = self.dataset.labelMapping(sem_label)
        mapped = torch.as_tensor(np.ascontiguousarray(mapped).astype(np.int32)) if not isinstance(mapped, torch.Tensor) \
            else mapped
        if self._mean_dev is None:
            self._mean_dev = self.proj_img_mean.to(self.device)
            self._stds_dev = self.proj_img_stds.to(self.device)
        feat, label, mask, rng = self.projection.loader_item(pts, mapped, self._mean_dev, self._stds_dev)
        if self.return_uproj:
            c = self.projection.cached_data
            return feat, label, mask, rng, c["uproj_x_idx"].long(), c["uproj_y_idx"].long(), c["uproj_depth"]
        return feat, label, mask

    def __len__(self):
        if 0 < self.data_len < len(self.dataset):
            return self.data_len
        return len(self.dataset)
