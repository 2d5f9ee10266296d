"""Trainer for tasks/epmf (counterpart of the reference's tasks/epmf/trainer.py:12-679, EPMFNet branch).

What differs from tasks/pmf, as in the reference: the single-tensor batches of PerspectiveViewLoaderV2 ([N,10,H,W]: channels
0-4 LiDAR, 5-7 RGB, 8 mask, 9 label, :358-371), class weights from the configured ``cls_freq`` (:119-139,243-251), the
``PVconfig`` normalisation constants (:352-355), AdamW with the configured weight decay over the LiDAR stream AND the
MultiTaskLoss sigmas (:95-109) and the six-term objective through MultiTaskLoss(6) when ``use_mtloss`` is set (:27-33,
409-430).  The per-iteration work is pmf_amd.engine.EPMFEngine (use_mtloss) or TrainEngine with the fixed lambda / gamma
weights; everything else -- loaders, prefetch thread, logging, metrics, return value of run() -- is the tasks/pmf trainer."""
import importlib.util
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

import pc_processor
from pmf_amd.engine import EPMFEngine, TrainEngine
from pmf_amd.loss import EPMF_TERMS
from pmf_amd.utils.detinit import synthetic_batch

_spec = importlib.util.spec_from_file_location(
    "pmf_task_trainer", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pmf", "trainer.py"))
base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(base)


class SyntheticPV2(Dataset):
    """[10,H,W] items shaped like PerspectiveViewLoaderV2's (perspective_view_loader_v2.py:146-157)."""

    def __init__(self, n, h, w, nclasses, seed=0):
        self.n, self.h, self.w, self.nclasses, self.seed = n, h, w, nclasses, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        pcd, rgb, label, mask = synthetic_batch(1, self.h, self.w, self.nclasses, seed=self.seed + i, fill=0.3)
        return torch.cat((pcd[0], rgb[0], mask[0][None], label[0].float()[None]), 0)


class Trainer(base.Trainer):
    TERMS = EPMF_TERMS
    TERM_TAGS = ("LossImageFocal", "LossImageLovasz", "LossImagePerception", "LossPerception", "LossFocal", "LossLovasz")

    def _initEngine(self, total):
        s = self.settings
        pv = s.config["PVconfig"]
        kw = dict(lr=s.lr, momentum=s.momentum, weight_decay=s.weight_decay, tau=s.tau, alpha=self.alpha,
                  ignore_class=self.ignore_class, warmup_steps=s.warmup_epochs * total,
                  max_steps=total * (s.n_epochs - s.warmup_epochs), feature_mean=pv["pcd_mean"], feature_std=pv["pcd_stds"],
                  distributed=s.distributed and s.world_size > 1, device_ids=[s.gpu] if s.distributed else None)
        if s.use_mtloss:
            self.TERMS, eng = EPMF_TERMS, EPMFEngine(self.model, s.nclasses, **kw)
            self.mt_loss = eng.mt_loss
            return eng
        # fixed weights (:403-404,414-415): the tasks/pmf objective, AdamW with the configured weight decay (:105-109)
        self.TERMS, self.TERM_TAGS = base.TERMS, base.Trainer.TERM_TAGS
        return TrainEngine(self.model, s.nclasses, lambda_=s.lambda_, gamma=s.gamma, adam_weight_decay=s.weight_decay, **kw)

    @staticmethod
    def _unpack(batch):
        x = batch.cuda(non_blocking=True)                    # trainer.py:358-371
        return x[:, 0:8], x[:, 8], x[:, 9]

    def _initDataloader(self):
        s = self.settings
        cls_freq = np.array(s.cls_freq, np.float64)
        cls_freq = cls_freq / cls_freq.sum()
        cls_freq[0] = 0
        pv = s.config["PVconfig"]
        device_side = True
        if s.dataset == "SemanticKitti":                                       # trainer.py:119-150
            cfg_path = s.config.get("data_config_path") or pc_processor.dataset.semantic_kitti.DEFAULT_CONFIG
            seqs = s.config.get("sequences", {})
            trainset = pc_processor.dataset.semantic_kitti.SemanticKitti(
                root=s.data_root, sequences=list(seqs.get("train", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10])), config_path=cfg_path)
            valset = pc_processor.dataset.semantic_kitti.SemanticKitti(
                root=s.data_root, sequences=list(seqs.get("valid", [8])), config_path=cfg_path)
            self.cls_weight = 1 / (cls_freq + 1e-8)
            self.cls_weight[0] = 0
            self.ignore_class = []
            li = trainset.data_config.get("learning_ignore", {})
            for cl in range(len(self.cls_weight)):
                if li.get(cl, False):
                    self.cls_weight[cl] = 0
                if self.cls_weight[cl] < 1e-10:
                    self.ignore_class.append(cl)
            self.mapped_cls_name = trainset.mapped_cls_name
        elif s.dataset == "nuScenes":                                          # trainer.py:152-167 (needs nuscenes-devkit)
            trainset = pc_processor.dataset.nuScenes.Nuscenes(root=s.data_root, version="v1.0-trainval", split="train")
            valset = pc_processor.dataset.nuScenes.Nuscenes(root=s.data_root, version="v1.0-trainval", split="val")
            self.cls_weight = 1 / (cls_freq + 1e-8)
            self.cls_weight[0] = 0
            self.ignore_class = [0]
            self.mapped_cls_name = trainset.mapped_cls_name
        elif s.dataset == "Synthetic":
            nfr = s.config.get("synthetic_frames", [16, 4])
            train_pv = SyntheticPV2(nfr[0], pv["proj_ht"], pv["proj_wt"], s.nclasses, seed=s.seed)
            val_pv = SyntheticPV2(nfr[1], pv["proj_h"], pv["proj_w"], s.nclasses, seed=s.seed + 10000)
            trainset, valset, device_side = train_pv, val_pv, False
            self.cls_weight = 1 / (cls_freq + 1e-8)
            self.cls_weight[0] = 0
            self.ignore_class = [0]
            self.mapped_cls_name = {i: "class_%d" % i for i in range(s.nclasses)}
        else:
            raise ValueError("invalid dataset: {}".format(s.dataset))
        alpha = np.log(1 + self.cls_weight)                                    # _initCriterion, trainer.py:243-251
        alpha = alpha / alpha.max()
        alpha[0] = 0
        self.alpha = alpha.astype(np.float32)
        if self.recorder is not None:
            self.recorder.logger.info("weight: {}".format(self.cls_weight))
            self.recorder.logger.info("focal_loss alpha: {}".format(self.alpha))
        if device_side:                                                        # trainer.py:196-204
            train_pv = pc_processor.dataset.PerspectiveViewLoaderV2(dataset=trainset, config=s.config, is_train=True, img_aug=True)
            val_pv = pc_processor.dataset.PerspectiveViewLoaderV2(dataset=valset, config=s.config, is_train=False, img_aug=False)
        tsamp = vsamp = None
        if s.distributed and s.world_size > 1:
            tsamp = torch.utils.data.distributed.DistributedSampler(train_pv, shuffle=True, drop_last=True)
            vsamp = torch.utils.data.distributed.DistributedSampler(val_pv, shuffle=False, drop_last=False)
        workers = 0 if device_side else s.n_threads
        tl = DataLoader(train_pv, batch_size=s.batch_size[0], num_workers=workers, shuffle=tsamp is None, sampler=tsamp,
                        drop_last=True)
        vl = DataLoader(val_pv, batch_size=s.batch_size[1], num_workers=workers, shuffle=False, sampler=vsamp, drop_last=False)
        if device_side and s.n_threads > 0:
            tl, vl = base.Prefetcher(tl, workers=s.n_threads), base.Prefetcher(vl, workers=s.n_threads)
        return tl, vl, tsamp, vsamp
