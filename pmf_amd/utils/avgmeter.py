"""AverageMeter / RemainTime (pc_processor/utils/avgmeter.py, utils.py) -- logging helpers of the trainer."""


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)


class RemainTime:
    def __init__(self, epoch):
        self.epoch = epoch
        self.timer_avg = {}
        self.total_iter = {}

    def update(self, cost_time, batch_size=1, mode="Train"):
        self.timer_avg.setdefault(mode, AverageMeter()).update(cost_time, batch_size)

    def reset(self):
        for v in self.timer_avg.values():
            v.reset()

    def getRemainTime(self, epoch, iters, total_iter, mode="Train"):
        self.total_iter[mode] = total_iter
        remain = 0.0
        for k, v in self.timer_avg.items():
            if k == mode:
                remain += ((self.epoch - epoch) * total_iter - iters) * v.avg
            else:
                remain += (self.epoch - epoch) * self.total_iter.get(k, 0) * v.avg
        return remain
