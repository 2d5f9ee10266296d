"""Confusion-matrix metrics -- call surface of pc_processor/metrics/iou_eval.py:9-104.

Differences that do not change the reported numbers:
  * the confusion matrix lives on the device of the predictions (bincount, no D2H copy of 2 MB argmax maps);
  * under DDP the reference does barrier + all_reduce inside EVERY getIoU/getAcc/getRecall call (6 pairs per
    training iteration, tasks/pmf/trainer.py:387-396).  Here ``getStats`` all-reduces at most once per update
    (cached until the next addBatch) and ``sync=False`` returns rank-local numbers; ``sync_every=1`` semantics
    reproduce the reference exactly, the trainer's default defers the reduction to its print frequency."""
import torch
import torch.distributed as dist


class IOUEval:
    def __init__(self, n_classes, device=torch.device("cpu"), ignore=None, is_distributed=False):
        self.n_classes = n_classes
        self.device = torch.device(device)
        ignore = [] if ignore is None else list(ignore)
        self.ignore = torch.tensor(ignore).long()
        self.include = torch.tensor([n for n in range(n_classes) if n not in ignore]).long()
        self.is_distributed = is_distributed
        self.reset()

    def num_classes(self):
        return self.n_classes

    def reset(self):
        self.conf_matrix = torch.zeros((self.n_classes, self.n_classes), dtype=torch.long, device=self.device)
        self._cache = None

    def addBatch(self, x, y):
        x = torch.as_tensor(x).reshape(-1).long()
        y = torch.as_tensor(y).reshape(-1).long().to(x.device)
        if self.conf_matrix.device != x.device:
            self.conf_matrix = self.conf_matrix.to(x.device)
        idx = x * self.n_classes + y                       # rows = prediction, cols = ground truth
        self.conf_matrix += torch.bincount(idx, minlength=self.n_classes ** 2).view(self.n_classes, self.n_classes)
        self._cache = None

    def external_update(self):
        """the fused GPU loss (pmf_loss_pixel) added a batch to conf_matrix in place: drop cached statistics."""
        self._cache = None

    def getStats(self, sync=True):
        if self._cache is not None and self._cache[0] == sync:
            return self._cache[1]
        conf = self.conf_matrix.clone().double()
        if sync and self.is_distributed and dist.is_available() and dist.is_initialized():
            dist.all_reduce(conf)
        if self.ignore.numel():
            ig = self.ignore.to(conf.device)
            conf[ig] = 0
            conf[:, ig] = 0
        tp = conf.diag()
        fp = conf.sum(dim=1) - tp
        fn = conf.sum(dim=0) - tp
        self._cache = (sync, (tp, fp, fn))
        return tp, fp, fn

    def getIoU(self, sync=True):
        tp, fp, fn = self.getStats(sync)
        iou = tp / (tp + fp + fn + 1e-15)
        return iou[self.include.to(iou.device)].mean(), iou

    def getAcc(self, sync=True):
        tp, fp, fn = self.getStats(sync)
        acc = tp / (tp + fp + 1e-15)
        return acc[self.include.to(acc.device)].mean(), acc

    def getRecall(self, sync=True):
        tp, fp, fn = self.getStats(sync)
        rec = tp / (tp + fn + 1e-15)
        return rec[self.include.to(rec.device)].mean(), rec
