"""Trainer for tasks/pmf (counterpart of the reference's tasks/pmf/trainer.py:13-537).

Same responsibilities -- data loaders, criterion, AdamW(lidar)+SGD(camera), WarmupCosineLR x2, DDP, two IOUEval,
run(epoch, "Train"|"Validation") returning the epoch summary -- with the per-iteration work delegated to
pmf_amd.engine.TrainEngine (the unit bench.py times).  dataset: "Synthetic" needs no files (no SemanticKITTI here);
"SemanticKitti" expects the reference's parser object to be importable by the user."""
import datetime
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

import pc_processor
from pmf_amd.engine import TrainEngine, kitti_focal_alpha
from pmf_amd.utils.detinit import synthetic_batch


class SyntheticPV(Dataset):
    """[8,H,W] feature / mask / label triples shaped like PerspectiveViewLoader's output (loader :138-141)."""

    def __init__(self, n, h, w, nclasses, seed=0):
        self.n, self.h, self.w, self.nclasses, self.seed = n, h, w, nclasses, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        pcd, rgb, label, mask = synthetic_batch(1, self.h, self.w, self.nclasses, seed=self.seed + i)
        return torch.cat((pcd[0], rgb[0]), 0), mask[0], label[0].float()


class Trainer(object):
    def __init__(self, settings, model, recorder=None):
        self.settings, self.recorder = settings, recorder
        self.model = model.cuda()
        self.remain_time = pc_processor.utils.RemainTime(settings.n_epochs)
        self.train_loader, self.val_loader, self.train_sampler, self.val_sampler = self._initDataloader()
        sensor = settings.config["sensor"]
        total = len(self.train_loader)
        self.engine = TrainEngine(
            self.model, settings.nclasses, lr=settings.lr, momentum=settings.momentum,
            weight_decay=settings.weight_decay, lambda_=settings.lambda_, gamma=settings.gamma, tau=settings.tau,
            alpha=self.alpha, ignore_class=self.ignore_class, warmup_steps=settings.warmup_epochs * total,
            max_steps=total * (settings.n_epochs - settings.warmup_epochs),
            feature_mean=sensor["img_mean"], feature_std=sensor["img_stds"],
            distributed=settings.distributed and settings.world_size > 1,
            device_ids=[settings.gpu] if settings.distributed else None)
        self.optimizer, self.aux_optimizer = self.engine.optimizer, self.engine.aux_optimizer
        self.metrics, self.metrics_img = self.engine.metrics, self.engine.metrics_img
        self.scheduler, self.aux_scheduler = self.engine.scheduler, self.engine.aux_scheduler

    def _initDataloader(self):
        s = self.settings
        sensor = s.config["sensor"]
        if s.dataset == "Synthetic":
            nfr = s.config.get("synthetic_frames", [16, 4])
            trainset = SyntheticPV(nfr[0], sensor["proj_ht"], sensor["proj_wt"], s.nclasses, seed=s.seed)
            valset = SyntheticPV(nfr[1], sensor["proj_h"], sensor["proj_w"], s.nclasses, seed=s.seed + 10000)
            self.alpha = np.ones(s.nclasses, np.float32)
            self.alpha[0] = 0
            self.ignore_class = [0]
            self.mapped_cls_name = ["class_%d" % i for i in range(s.nclasses)]
        else:
            raise NotImplementedError("dataset {}: plug the reference's SemanticKitti parser into "
                                      "pmf_amd.dataset.PerspectiveViewLoader (INTEGRATION.md)".format(s.dataset))
        tsamp = vsamp = None
        if s.distributed and s.world_size > 1:
            tsamp = torch.utils.data.distributed.DistributedSampler(trainset, shuffle=True, drop_last=True)
            vsamp = torch.utils.data.distributed.DistributedSampler(valset, shuffle=False, drop_last=False)
        tl = DataLoader(trainset, batch_size=s.batch_size[0], num_workers=s.n_threads, shuffle=tsamp is None,
                        sampler=tsamp, drop_last=True)
        vl = DataLoader(valset, batch_size=s.batch_size[1], num_workers=s.n_threads, shuffle=False, sampler=vsamp)
        return tl, vl, tsamp, vsamp

    def run(self, epoch, mode="Train"):
        s, eng = self.settings, self.engine
        if mode == "Train":
            loader = self.train_loader
            if self.train_sampler is not None:
                self.train_sampler.set_epoch(epoch)
        elif mode == "Validation":
            loader = self.val_loader
        else:
            raise ValueError("invalid mode: {}".format(mode))
        self.metrics.reset()
        self.metrics_img.reset()
        meters = {k: pc_processor.utils.AverageMeter() for k in ("loss", "foc", "lov", "foc_cam", "lov_cam", "per")}
        total_iter = len(loader)
        t_start = time.time()
        for i, (feat, mask, label) in enumerate(loader):
            t0 = time.time()
            feat, mask, label = feat.cuda(non_blocking=True), mask.cuda(non_blocking=True), label.cuda(non_blocking=True)
            step = eng.train_step if mode == "Train" else eng.eval_step
            total, terms = step(feat, mask, label)
            last = (i + 1) % max(s.print_frequency, 1) == 0 or i + 1 == total_iter
            if last:                                      # host syncs only when something is printed
                vals = torch.stack([total] + [terms[k] for k in ("foc", "lov", "foc_cam", "lov_cam", "per")]).tolist()
                for k, v in zip(meters, vals):
                    meters[k].update(v, feat.size(0))
                miou, _ = self.metrics.getIoU()
                macc, _ = self.metrics.getAcc()
                mrec, _ = self.metrics.getRecall()
                miou_i, _ = self.metrics_img.getIoU()
                self.remain_time.update(cost_time=(time.time() - t_start), mode=mode)
                rt = datetime.timedelta(seconds=int(self.remain_time.getRemainTime(epoch, i, total_iter, mode)))
                if self.recorder is not None:
                    self.recorder.logger.info(
                        ">>> {} E[{:03d}|{:03d}] I[{:04d}|{:04d}] DT[{:.3f}] PT[{:.3f}] LR {:0.5f} Loss {:0.4f} Acc {:0.4f} "
                        "IOU {:0.4f} Recall {:0.4f} ImgIOU {:0.4f} RT {}".format(
                            mode, s.n_epochs, epoch + 1, total_iter, i + 1, t0 - t_start, time.time() - t0,
                            self.optimizer.param_groups[0]["lr"], meters["loss"].avg, macc.item(), miou.item(),
                            mrec.item(), miou_i.item(), rt))
            t_start = time.time()
            if s.is_debug:
                break
        miou, ciou = self.metrics.getIoU()
        macc, _ = self.metrics.getAcc()
        mrec, _ = self.metrics.getRecall()
        if self.recorder is not None:
            for k, m in meters.items():
                self.recorder.tensorboard.add_scalar("{}_{}".format(mode, k), m.avg, epoch)
            self.recorder.tensorboard.add_scalar("{}_IOU".format(mode), miou.item(), epoch)
        return {"Loss": meters["loss"].avg, "Acc": macc.item(), "IOU": miou.item(), "Recall": mrec.item(),
                "class_IOU": ciou.tolist()}
