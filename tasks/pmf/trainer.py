"""Trainer for tasks/pmf (counterpart of the reference's tasks/pmf/trainer.py:13-537).

Same responsibilities and the same constructor / ``run(epoch, mode)`` contract: data loaders (SemanticKitti / nuScenes
through PerspectiveViewLoader exactly as trainer.py:100-147 builds them, plus a file-free "Synthetic" set), class weights
-> focal alpha (:108-114,194-199), AdamW(lidar) + SGD-Nesterov(camera) (:80-98), two WarmupCosineLR (:61-75), data
parallelism (:33-39), two IOUEval (:49-59); ``run`` returns {"Acc", "IOU", "Recall", "last"} (:531-537).  The
per-iteration work (:289-341) is pmf_amd.engine.TrainEngine -- the unit bench.py times.

Differences that do not change results:
  * the perspective loaders run HIP kernels and return DEVICE tensors, so their DataLoader runs in the main process
    (num_workers=0: a forked worker cannot initialise the GPU); file reading / PNG decoding of the next batches is
    overlapped by a small prefetch thread instead (``n_threads`` > 0 switches it on);
  * loss terms are accumulated on the device every iteration and read at the print frequency (one host sync per
    print instead of nine .item() calls per iteration); metrics are all-reduced when read."""
import datetime
import os
import queue
import threading
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

import pc_processor
from pmf_amd.engine import TrainEngine, kitti_focal_alpha
from pmf_amd.utils.detinit import synthetic_batch

TERMS = ("foc", "lov", "foc_cam", "lov_cam", "per")


class SyntheticPV(Dataset):
    """[8,H,W] feature / mask / label triples shaped like PerspectiveViewLoader's output (loader :138-141)."""

    def __init__(self, n, h, w, nclasses, seed=0):
        self.n, self.h, self.w, self.nclasses, self.seed = n, h, w, nclasses, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        pcd, rgb, label, mask = synthetic_batch(1, self.h, self.w, self.nclasses, seed=self.seed + i)
        return torch.cat((pcd[0], rgb[0]), 0), mask[0], label[0].float()


def _record_stream(x, stream):
    if torch.is_tensor(x):
        if x.is_cuda:
            x.record_stream(stream)
    elif isinstance(x, (list, tuple)):
        for y in x:
            _record_stream(y, stream)
    elif isinstance(x, dict):
        for y in x.values():
            _record_stream(y, stream)


class Prefetcher(object):
    """iterates a DataLoader from background threads, ``depth`` batches ahead per thread, each on a HIP stream of its own.
    The device-side loaders upload a frame with a pageable host -> device copy; on the training stream it would wait for
    everything queued there (a whole step), which serialises the thread behind the GPU (measured: no gain at all).  On a
    side stream the uploads and the loader kernels run under the step; every batch carries an event the consumer's stream
    waits on, and its tensors are marked as used by that stream (caching-allocator safety).
    ``workers`` > 1 (a plain DataLoader with num_workers=0 only): worker j builds batches j, j + workers, ... of the
    loader's batch sampler itself (dataset items + collate), so PNG decoding and the uploads of several batches overlap;
    batches are delivered in sampler order, but the workers draw their augmentation parameters from the global RNGs
    concurrently -- like the reference's multi-process DataLoader the draws are then not reproducible run to run.
    The producers stop when the consumer stops: leaving the loop early (debug break, exception, generator collected) sets
    a flag they check between items, so no thread -- and none of the device batches it holds -- outlives an epoch."""

    MAX_WORKERS = 3      # the reference's n_threads (DataLoader processes, 4-8) would mean that many device-resident batches

    def __init__(self, loader, depth=2, workers=1):
        self.loader = loader
        workers = min(int(workers), self.MAX_WORKERS)
        self.workers = workers if (workers > 1 and isinstance(loader, DataLoader) and loader.num_workers == 0
                                   and loader.batch_sampler is not None) else 1
        # ``depth`` batches ahead IN TOTAL (not per thread): at least one slot per worker
        self.depth = max(1, -(-int(depth) // self.workers))

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        W = self.workers
        qs = [queue.Queue(maxsize=self.depth) for _ in range(W)]
        stop = threading.Event()
        gpu = torch.cuda.is_available()
        dev = torch.cuda.current_device() if gpu else None
        batches = list(self.loader.batch_sampler) if W > 1 else None      # the index lists, drawn once, in order

        def put(q, x):                            # False: the consumer is gone
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def source(j):
            if W == 1:
                return iter(self.loader)
            ds, collate = self.loader.dataset, self.loader.collate_fn
            return (collate([ds[i] for i in idx]) for idx in batches[j::W])

        def work(j):
            q = qs[j]
            try:
                if gpu:
                    torch.cuda.set_device(dev)
                    side = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(side):
                        for item in source(j):
                            ev = torch.cuda.Event()
                            ev.record(side)
                            if not put(q, ("item", (item, ev))):
                                return
                else:
                    for item in source(j):
                        if not put(q, ("item", (item, None))):
                            return
                put(q, ("end", None))
            except BaseException as e:            # surfaces in the consumer
                put(q, ("error", e))
        ths = [threading.Thread(target=work, args=(j,), daemon=True) for j in range(W)]
        for th in ths:
            th.start()
        try:
            j = 0
            while True:
                kind, payload = qs[j].get()
                if kind == "end":                 # worker j is the first to run out (sampler order): all are done
                    return
                if kind == "error":
                    raise payload
                item, ev = payload
                if ev is not None:
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(ev)
                    _record_stream(item, cur)
                yield item
                j = (j + 1) % W
        finally:
            stop.set()
            for q in qs:                          # release what the producers queued, then let them finish
                while True:
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        break
            for th in ths:
                th.join(timeout=10.0)


class Trainer(object):
    def __init__(self, settings, model, recorder=None):
        self.settings, self.recorder = settings, recorder
        self.model = model.cuda()
        self.remain_time = pc_processor.utils.RemainTime(settings.n_epochs)
        self.train_loader, self.val_loader, self.train_sampler, self.val_sampler = self._initDataloader()
        self.engine = self._initEngine(len(self.train_loader))
        # main.py reads / restores these two (:72-83,104-127): reference checkpoint layout on top of the flat state
        self.optimizer, self.aux_optimizer = self.engine.optimizer_view, self.engine.aux_optimizer_view
        self.metrics, self.metrics_img = self.engine.metrics, self.engine.metrics_img
        self.scheduler, self.aux_scheduler = self.engine.scheduler, self.engine.aux_scheduler
        self.last_summary = {}

    TERMS = TERMS                       # loss terms the engine reports, in logging order
    TERM_TAGS = ("LossFocal", "LossLovasz", "LossImageFocal", "LossImageLovasz", "LossPerception")

    def _initEngine(self, total):
        """the per-iteration work (trainer.py:289-341); tasks/epmf overrides this with the six-term engine"""
        s = self.settings
        sensor = s.config["sensor"]
        return TrainEngine(
            self.model, s.nclasses, lr=s.lr, momentum=s.momentum, weight_decay=s.weight_decay, lambda_=s.lambda_,
            gamma=s.gamma, tau=s.tau, alpha=self.alpha, ignore_class=self.ignore_class,
            warmup_steps=s.warmup_epochs * total, max_steps=total * (s.n_epochs - s.warmup_epochs),
            feature_mean=sensor["img_mean"], feature_std=sensor["img_stds"],
            distributed=s.distributed and s.world_size > 1, device_ids=[s.gpu] if s.distributed else None)

    @staticmethod
    def _unpack(batch):
        """one DataLoader batch -> (feature [N,8,H,W], mask [N,H,W], label [N,H,W]) (trainer.py:289-297)"""
        feat, mask, label = batch
        return feat.cuda(non_blocking=True), mask.cuda(non_blocking=True), label.cuda(non_blocking=True)

    # ------------------------------------------------------------------ data
    def _initDataloader(self):
        s = self.settings
        sensor = s.config["sensor"]
        device_side = True
        if s.dataset == "SemanticKitti":                                       # trainer.py:101-125
            cfg_path = s.config.get("data_config_path") or pc_processor.dataset.semantic_kitti.DEFAULT_CONFIG
            seqs = s.config.get("sequences", {})
            trainset = pc_processor.dataset.semantic_kitti.SemanticKitti(
                root=s.data_root, sequences=list(seqs.get("train", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10])), config_path=cfg_path)
            valset = pc_processor.dataset.semantic_kitti.SemanticKitti(
                root=s.data_root, sequences=list(seqs.get("valid", [8])), config_path=cfg_path)
            li = trainset.data_config.get("learning_ignore", {})
            ignore = [bool(li.get(c, False)) for c in range(len(trainset.cls_freq))]
            self.alpha, self.ignore_class = kitti_focal_alpha(trainset.cls_freq, ignore)
            self.cls_weight = np.where(np.asarray(ignore), 0.0, 1.0 / (trainset.cls_freq + 1e-3))
            if self.recorder is not None:
                self.recorder.logger.info("weight: {}".format(self.cls_weight))
                self.recorder.logger.info("focal_loss alpha: {}".format(self.alpha))
            self.mapped_cls_name = trainset.mapped_cls_name
        elif s.dataset == "nuScenes":                                          # trainer.py:127-136
            trainset = pc_processor.dataset.nuScenes.Nuscenes(root=s.data_root, version="v1.0-trainval", split="train")
            valset = pc_processor.dataset.nuScenes.Nuscenes(root=s.data_root, version="v1.0-trainval", split="val")
            self.cls_weight = np.ones((s.nclasses))
            self.alpha = np.ones(s.nclasses, np.float32)
            self.alpha[0] = 0
            self.ignore_class = [0]
            self.mapped_cls_name = trainset.mapped_cls_name
        elif s.dataset == "Synthetic":
            nfr = s.config.get("synthetic_frames", [16, 4])
            train_pv = SyntheticPV(nfr[0], sensor["proj_ht"], sensor["proj_wt"], s.nclasses, seed=s.seed)
            val_pv = SyntheticPV(nfr[1], sensor["proj_h"], sensor["proj_w"], s.nclasses, seed=s.seed + 10000)
            trainset, valset, device_side = train_pv, val_pv, False
            self.alpha = np.ones(s.nclasses, np.float32)
            self.alpha[0] = 0
            self.ignore_class = [0]
            self.mapped_cls_name = {i: "class_%d" % i for i in range(s.nclasses)}
        else:
            raise ValueError("invalid dataset: {}".format(s.dataset))
        if device_side:                                                        # trainer.py:139-147, verbatim arguments
            train_pv = pc_processor.dataset.PerspectiveViewLoader(
                dataset=trainset, config=s.config, is_train=True, pcd_aug=False, img_aug=True, use_padding=True)
            val_pv = pc_processor.dataset.PerspectiveViewLoader(
                dataset=valset, config=s.config, is_train=False, use_padding=True)
        tsamp = vsamp = None
        if s.distributed and s.world_size > 1:
            tsamp = torch.utils.data.distributed.DistributedSampler(trainset, shuffle=True, drop_last=True)
            vsamp = torch.utils.data.distributed.DistributedSampler(valset, shuffle=False, drop_last=False)
        workers = 0 if device_side else s.n_threads
        tl = DataLoader(train_pv, batch_size=s.batch_size[0], num_workers=workers, shuffle=tsamp is None, sampler=tsamp,
                        drop_last=True)
        vl = DataLoader(val_pv, batch_size=s.batch_size[1], num_workers=workers, shuffle=False, sampler=vsamp,
                        drop_last=False)
        if device_side and s.n_threads > 0:      # n_threads: the reference's DataLoader workers = prefetch threads here
            tl, vl = Prefetcher(tl, workers=s.n_threads), Prefetcher(vl, workers=s.n_threads)
        return tl, vl, tsamp, vsamp

    # ------------------------------------------------------------------ one epoch
    def run(self, epoch, mode="Train"):
        s, eng = self.settings, self.engine
        if mode == "Train":
            loader = self.train_loader
            if self.train_sampler is not None:
                self.train_sampler.set_epoch(epoch)
        elif mode == "Validation":
            loader = self.val_loader     # (the engine broadcasts rank 0's BN statistics before the first eval step)
        else:
            raise ValueError("invalid mode: {}".format(mode))
        self.metrics.reset()
        self.metrics_img.reset()
        TERMS = self.TERMS
        sums = torch.zeros(1 + len(TERMS), dtype=torch.float64, device="cuda")     # sum of (loss x batch) per term
        count = 0
        total_iter = len(loader)
        t_start = time.time()
        lr = self.optimizer.param_groups[0]["lr"]
        for i, batch in enumerate(loader):
            t0 = time.time()
            feat, mask, label = self._unpack(batch)
            step = eng.train_step if mode == "Train" else eng.eval_step
            total, terms = step(feat, mask, label)
            sums += torch.stack([total.detach()] + [terms[k].detach() for k in TERMS]).double() * feat.size(0)
            count += feat.size(0)
            if (i + 1) % max(s.print_frequency, 1) == 0 or i + 1 == total_iter:    # the only host syncs of the loop
                avg = (sums / count).tolist()
                miou, _ = self.metrics.getIoU()
                macc, _ = self.metrics.getAcc()
                mrec, _ = self.metrics.getRecall()
                miou_i, _ = self.metrics_img.getIoU()
                self.remain_time.update(cost_time=(time.time() - t_start), mode=mode)
                rt = datetime.timedelta(seconds=int(self.remain_time.getRemainTime(epoch, i, total_iter, mode)))
                lr = self.optimizer.param_groups[0]["lr"]
                if self.recorder is not None:
                    self.recorder.logger.info(
                        ">>> {} E[{:03d}|{:03d}] I[{:04d}|{:04d}] DT[{:.3f}] PT[{:.3f}] LR {:0.5f} Loss {:0.4f} Acc {:0.4f} "
                        "IOU {:0.4f} Recall {:0.4f} ImgIOU {:0.4f} RT {}".format(
                            mode, s.n_epochs, epoch + 1, total_iter, i + 1, t0 - t_start, time.time() - t0, lr, avg[0],
                            macc.item(), miou.item(), mrec.item(), miou_i.item(), rt))
            t_start = time.time()
            if s.is_debug:
                break
        avg = (sums / max(count, 1)).tolist()
        miou, ciou = self.metrics.getIoU()
        macc, cacc = self.metrics.getAcc()
        mrec, crec = self.metrics.getRecall()
        miou_i, _ = self.metrics_img.getIoU()
        if self.recorder is not None:
            tb = self.recorder.tensorboard
            for k, v in zip(("Loss",) + tuple(self.TERM_TAGS), avg):
                tb.add_scalar("{}_{}".format(mode, k), v, epoch)
            tb.add_scalar("{}_lr".format(mode), lr, epoch)
            for k, v in (("meanAcc", macc), ("meanIOU", miou), ("meanRecall", mrec), ("Image_meanIOU", miou_i)):
                tb.add_scalar("{}_{}".format(mode, k), v.item(), epoch)
            for c, name in self.mapped_cls_name.items():
                tb.add_scalar("{}_{:02d}_{}_IOU".format(mode, c, name), ciou[c].item(), epoch)
            self.recorder.logger.info(">>> {} Loss {:0.4f} Acc {:0.4f} IOU {:0.4f} Recall {:0.4f}".format(
                mode, avg[0], macc.item(), miou.item(), mrec.item()))
        self.last_summary = {"Loss": avg[0], "terms": dict(zip(TERMS, avg[1:])), "class_IOU": ciou.tolist(),
                             "ImgIOU": miou_i.item()}
        return {"Acc": macc.item(), "IOU": miou.item(), "Recall": mrec.item(), "last": 0}
