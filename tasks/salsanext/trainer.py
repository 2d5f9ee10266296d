"""Trainer for tasks/salsanext -- the LiDAR-only range-image task (counterpart of the reference's
tasks/salsanext/trainer.py:14-324).

Same constructor / ``run(epoch, mode)`` contract and return value ({"Acc", "IOU", "Recall"}, :318-324): SemanticKitti /
nuScenes through SalsaNextLoader exactly as :60-128 builds them (plus a file-free "Synthetic" set of LiDAR sweeps), class
weights -> focal alpha (:86-95,152-160), AdamW over the whole model (:54-58), WarmupCosineLR (:46-51), one IOUEval.  The
per-iteration work (:186-212) is pmf_amd.engine.SalsaNextEngine: label / mask clean-up, SalsaNext on the HIP plan,
Lovasz + masked focal loss, AdamW on the flat state.  The range-image loader runs HIP kernels and returns device tensors,
so its DataLoader stays in the main process (``n_threads`` > 0 switches the prefetch thread of tasks/pmf on); losses are
accumulated on the device and read at the print frequency."""
import datetime
import importlib.util
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

import pc_processor
from pmf_amd.engine import SalsaNextEngine

_spec = importlib.util.spec_from_file_location(
    "pmf_task_trainer", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pmf", "trainer.py"))
base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(base)


class SyntheticSweeps(object):
    """the dataset duck type SalsaNextLoader reads (loadDataByIndex / labelMapping / __len__) over synthetic spinning-LiDAR
    sweeps: rings of points with range-dependent noise, labels by azimuth sector (no files)."""

    def __init__(self, n, n_classes, npts=40000, seed=0):
        self.n, self.n_classes, self.npts, self.seed = n, n_classes, npts, seed
        self.mapped_cls_name = {i: "class_%d" % i for i in range(n_classes)}

    def __len__(self):
        return self.n

    def loadDataByIndex(self, index):
        rng = np.random.Generator(np.random.PCG64(self.seed + index))
        az = rng.uniform(-np.pi, np.pi, self.npts)
        el = np.deg2rad(rng.uniform(-25.0, 3.0, self.npts))
        r = 2.0 + 60.0 * rng.random(self.npts) ** 2
        pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el), rng.random(self.npts)],
                       1).astype(np.float32)
        sem = (np.floor((az + np.pi) / (2 * np.pi) * (self.n_classes - 1)).astype(np.int32) % (self.n_classes - 1)) + 1
        sem[rng.random(self.npts) < 0.05] = 0
        return pts, sem, np.zeros_like(sem)

    def labelMapping(self, label):
        return label


class Trainer(object):
    def __init__(self, settings, model, recorder=None):
        self.settings, self.recorder = settings, recorder
        self.model = model.cuda()
        self.remain_time = pc_processor.utils.RemainTime(settings.n_epochs)
        self.train_loader, self.val_loader, self.train_sampler, self.val_sampler = self._initDataloader()
        total = len(self.train_loader)
        alpha = np.log(1 + self.cls_weight)                                   # _initCriterion, trainer.py:152-160
        alpha = alpha / alpha.max()
        alpha[0] = 0
        if self.recorder is not None:
            self.recorder.logger.info("focal_loss alpha: {}".format(alpha))
        self.engine = SalsaNextEngine(
            self.model, settings.n_classes, lr=settings.lr, momentum=settings.momentum, alpha=alpha.astype(np.float32),
            ignore_class=self.ignore_class, warmup_steps=settings.warmup_epochs * total,
            max_steps=total * (settings.n_epochs - settings.warmup_epochs),
            distributed=settings.distributed and settings.world_size > 1)
        # main.py saves / restores this one: the reference's per-parameter checkpoint layout on top of the flat state
        self.optimizer, self.scheduler, self.metrics = self.engine.optimizer_view, self.engine.scheduler, self.engine.metrics

    def _initDataloader(self):
        s = self.settings
        if s.dataset == "nuScenes":                                            # trainer.py:61-72 (needs nuscenes-devkit)
            trainset = pc_processor.dataset.nuScenes.Nuscenes(root=s.data_root, version="v1.0-trainval", split="train",
                                                              return_ref=False, has_image=False)
            valset = pc_processor.dataset.nuScenes.Nuscenes(root=s.data_root, version="v1.0-trainval", split="val",
                                                            return_ref=False, has_image=False)
            self.mapped_cls_name = trainset.mapped_cls_name
            self.ignore_class = [0]
            self.cls_weight = np.ones((s.n_classes))
            self.cls_weight[0] = 0
        elif s.dataset == "SemanticKitti":                                     # trainer.py:74-104
            cfg_path = s.config.get("data_config_path") or pc_processor.dataset.semantic_kitti.DEFAULT_CONFIG
            seqs = s.config.get("sequences", {})
            trainset = pc_processor.dataset.semantic_kitti.SemanticKitti(
                root=s.data_root, sequences=list(seqs.get("train", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10])), config_path=cfg_path,
                has_image=False)
            valset = pc_processor.dataset.semantic_kitti.SemanticKitti(
                root=s.data_root, sequences=list(seqs.get("valid", [8])), config_path=cfg_path, has_image=False)
            self.cls_weight = 1 / (trainset.cls_freq + 1e-3)
            self.ignore_class = []
            li = trainset.data_config.get("learning_ignore", {})
            for cl in range(len(self.cls_weight)):
                if li.get(cl, False):
                    self.cls_weight[cl] = 0
                if self.cls_weight[cl] < 1e-10:
                    self.ignore_class.append(cl)
            if self.recorder is not None:
                self.recorder.logger.info("weight: {}".format(self.cls_weight))
            self.mapped_cls_name = trainset.mapped_cls_name
        elif s.dataset == "Synthetic":
            nfr = s.config.get("synthetic_frames", [16, 4])
            trainset = SyntheticSweeps(nfr[0], s.n_classes, seed=s.seed)
            valset = SyntheticSweeps(nfr[1], s.n_classes, seed=s.seed + 10000)
            self.mapped_cls_name = trainset.mapped_cls_name
            self.ignore_class = [0]
            self.cls_weight = np.ones((s.n_classes))
            self.cls_weight[0] = 0
        else:
            raise ValueError("invalid dataset: {}".format(s.dataset))
        train_l = pc_processor.dataset.SalsaNextLoader(dataset=trainset, config=s.config)                     # :106-113
        val_l = pc_processor.dataset.SalsaNextLoader(dataset=valset, config=s.config, is_train=False)
        tsamp = vsamp = None
        if s.distributed and s.world_size > 1:
            tsamp = torch.utils.data.distributed.DistributedSampler(train_l, shuffle=True, drop_last=True)
            vsamp = torch.utils.data.distributed.DistributedSampler(val_l, shuffle=False, drop_last=False)
        tl = DataLoader(train_l, batch_size=s.batch_size, num_workers=0, shuffle=tsamp is None, sampler=tsamp, drop_last=True)
        vl = DataLoader(val_l, batch_size=s.batch_size, num_workers=0, shuffle=False, sampler=vsamp, drop_last=False)
        if s.n_threads > 0:
            tl, vl = base.Prefetcher(tl, workers=s.n_threads), base.Prefetcher(vl, workers=s.n_threads)
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
        sums = torch.zeros(3, dtype=torch.float64, device="cuda")             # loss, focal, lovasz (x batch size)
        count, total_iter, t_start = 0, len(loader), time.time()
        lr = self.optimizer.param_groups[0]["lr"]
        for i, (feat, label, mask) in enumerate(loader):                        # trainer.py:186: (feature, label, mask)
            t0 = time.time()
            feat, label, mask = feat.cuda(non_blocking=True), label.cuda(non_blocking=True), mask.cuda(non_blocking=True)
            step = eng.train_step if mode == "Train" else eng.eval_step
            total, terms = step(feat, label, mask)
            sums += torch.stack([total.detach(), terms["focal"], terms["lovasz"]]).double() * feat.size(0)
            count += feat.size(0)
            if (i + 1) % max(s.print_frequency, 1) == 0 or i + 1 == total_iter:
                avg = (sums / count).tolist()
                macc, _ = self.metrics.getAcc()
                miou, _ = self.metrics.getIoU()
                self.remain_time.update(cost_time=(time.time() - t_start), mode=mode)
                rt = datetime.timedelta(seconds=int(self.remain_time.getRemainTime(epoch, i, total_iter, mode)))
                lr = self.optimizer.param_groups[0]["lr"]
                if self.recorder is not None:
                    self.recorder.logger.info(
                        ">>> {} E[{:03d}|{:03d}] I[{:04d}|{:04d}] DT[{:.3f}] PT[{:.3f}] LR {} Loss {:0.4f} Acc {:0.4f} "
                        "IOU {:0.4F} RT {}".format(mode, s.n_epochs, epoch + 1, total_iter, i, t0 - t_start,
                                                   time.time() - t0, lr, avg[0], macc.item(), miou.item(), rt))
            t_start = time.time()
            if s.is_debug:
                break
        avg = (sums / max(count, 1)).tolist()
        macc, cacc = self.metrics.getAcc()
        miou, ciou = self.metrics.getIoU()
        mrec, crec = self.metrics.getRecall()
        if self.recorder is not None:
            tb = self.recorder.tensorboard
            for k, v in zip(("Loss", "LossSoftmax", "LossLovasz"), avg):
                tb.add_scalar("{}_{}".format(mode, k), v, epoch)
            for k, v in (("meanAcc", macc), ("meanIOU", miou), ("meanRecall", mrec)):
                tb.add_scalar("{}_{}".format(mode, k), v.item(), epoch)
            tb.add_scalar("{}_lr".format(mode), lr, epoch)
            for i, (_, name) in enumerate(self.mapped_cls_name.items()):
                tb.add_scalar("{}_{:02d}_{}_Acc".format(mode, i, name), cacc[i].item(), epoch)
                tb.add_scalar("{}_{:02d}_{}_Recall".format(mode, i, name), crec[i].item(), epoch)
                tb.add_scalar("{}_{:02d}_{}_IOU".format(mode, i, name), ciou[i].item(), epoch)
            self.recorder.logger.info(">>> {} Loss {:0.4f} Acc {:0.4f} IOU {:0.4F} Recall {:0.4f}".format(
                mode, avg[0], macc.item(), miou.item(), mrec.item()))
        self.last_summary = {"Loss": avg[0], "LossSoftmax": avg[1], "LossLovasz": avg[2]}
        return {"Acc": macc.item(), "IOU": miou.item(), "Recall": mrec.item()}
