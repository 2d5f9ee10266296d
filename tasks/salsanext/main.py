"""Entry point of tasks/salsanext (counterpart of the reference's tasks/salsanext/main.py:12-126): distributed init, seeds,
Recorder on the main process, SalsaNext(in_channels 5), Trainer, pretrained-weight / checkpoint restore, epoch loop with the
validation schedule, best_{Acc,IOU,Recall}_model.pth and checkpoint.pth ({"model", "optimizer", "epoch"}).

    python main.py config_synthetic.yaml
"""
import argparse
import datetime
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pc_processor  # noqa: E402
import trainer  # noqa: E402
from option import Option  # noqa: E402


class Experiment(object):
    def __init__(self, settings):
        self.settings = settings
        pc_processor.utils.init_distributed_mode(settings)
        if settings.distributed:
            torch.distributed.barrier()
        torch.manual_seed(settings.seed)
        torch.cuda.manual_seed(settings.seed)
        torch.cuda.set_device(settings.gpu if isinstance(settings.gpu, int) else 0)
        self.recorder = None
        if not settings.distributed or settings.rank == 0:
            settings.check_path()
            self.recorder = pc_processor.checkpoint.Recorder(settings, settings.save_path)
        self.epoch_start = 0
        if settings.net_type == "SalsaNext":                                   # main.py:37-40
            self.model = pc_processor.models.SalsaNext(in_channels=5, nclasses=settings.n_classes)
        else:
            raise ValueError("invalid netType: {}".format(settings.net_type))
        self._loadPretrained()
        self.trainer = trainer.Trainer(settings, self.model, self.recorder)
        self._loadCheckpoint()

    def _loadPretrained(self):
        s = self.settings
        assert s.pretrained_model is None or s.checkpoint is None, \
            "cannot use pretrained weight and checkpoint at the same time"
        if s.pretrained_model is None:
            return
        if not os.path.isfile(s.pretrained_model):
            raise FileNotFoundError("pretrained model not found: {}".format(s.pretrained_model))
        self.model.load_state_dict(torch.load(s.pretrained_model, map_location="cpu"))
        if self.recorder is not None:
            self.recorder.logger.info("loading pretrained weight from: {}".format(s.pretrained_model))

    def _loadCheckpoint(self):
        s = self.settings
        if s.checkpoint is None:
            return
        if not os.path.isfile(s.checkpoint):
            raise FileNotFoundError("checkpoint file not found: {}".format(s.checkpoint))
        ck = torch.load(s.checkpoint, map_location="cpu")
        self.model.load_state_dict(ck["model"])
        self.trainer.optimizer.load_state_dict(ck["optimizer"])
        self.epoch_start = ck["epoch"] + 1

    def run(self):
        s = self.settings
        t_start = time.time()
        if s.val_only:
            self.trainer.run(0, mode="Validation")
            return
        best = None
        for epoch in range(self.epoch_start, s.n_epochs):
            self.trainer.run(epoch, mode="Train")
            if epoch % s.val_frequency == 0 or epoch == s.n_epochs - 1:
                res = self.trainer.run(epoch, mode="Validation")
                if self.recorder is not None:
                    if best is None:
                        best = dict(res)
                    for k, v in res.items():
                        if v >= best[k]:
                            self.recorder.logger.info("get better {} model: {}".format(k, v))
                            best[k] = v
                            torch.save(self.model.state_dict(),
                                       os.path.join(self.recorder.checkpoint_path, "best_{}_model.pth".format(k)))
            if self.recorder is not None:
                torch.save({"model": self.model.state_dict(), "optimizer": self.trainer.optimizer.state_dict(), "epoch": epoch},
                           os.path.join(self.recorder.checkpoint_path, "checkpoint.pth"))
                if best is not None:
                    self.recorder.logger.info(">>> Best Result: " + " ".join("{}: {}".format(k, v) for k, v in best.items()))
            if s.is_debug:
                break
        if self.recorder is not None:
            self.recorder.logger.info("==== total cost time: {}".format(datetime.timedelta(seconds=time.time() - t_start)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="SalsaNext (LiDAR-only) training on MI355X")
    ap.add_argument("config_path", type=str, metavar="config_path")
    ap.add_argument("--id", type=int, default=0)
    a = ap.parse_args()
    exp = Experiment(Option(a.config_path))
    print("===init env success===")
    exp.run()
