"""Entry point (counterpart of the reference's tasks/pmf/main.py:11-149).

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 main.py config_synthetic.yaml
    python main.py config_synthetic.yaml                      # single GPU
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pc_processor  # noqa: E402
import trainer  # noqa: E402
from option import Option  # noqa: E402


class Experiment(object):
    def __init__(self, settings):
        self.settings = settings
        pc_processor.utils.init_distributed_mode(settings)
        torch.manual_seed(settings.seed)                       # same seed on every rank (main.py:20-21)
        torch.cuda.manual_seed(settings.seed)
        torch.cuda.set_device(settings.gpu if isinstance(settings.gpu, int) else 0)
        self.recorder = None
        if pc_processor.utils.is_main_process():
            self.recorder = pc_processor.checkpoint.Recorder(settings, settings.save_path)
        self.epoch_start = 0
        self.model = pc_processor.models.PMFNet(
            pcd_channels=5, img_channels=3, nclasses=settings.nclasses, base_channels=settings.base_channels,
            image_backbone=settings.img_backbone, imagenet_pretrained=settings.imagenet_pretrained)
        self.trainer = trainer.Trainer(settings, self.model, self.recorder)
        self._loadCheckpoint()

    def _loadCheckpoint(self):
        s = self.settings
        assert s.pretrained_model is None or s.checkpoint is None, \
            "cannot use pretrained weight and checkpoint at the same time"
        if s.pretrained_model is not None:
            if not os.path.isfile(s.pretrained_model):
                raise FileNotFoundError("pretrained model not found: {}".format(s.pretrained_model))
            sd = torch.load(s.pretrained_model, map_location="cpu")
            own = self.model.state_dict()
            own.update({k: v for k, v in sd.items() if k in own and own[k].size() == v.size()})   # shape-filtered
            self.model.load_state_dict(own)
        if s.checkpoint is not None:
            if not os.path.isfile(s.checkpoint):
                raise FileNotFoundError("checkpoint file not found: {}".format(s.checkpoint))
            ck = torch.load(s.checkpoint, map_location="cpu")
            self.model.load_state_dict(ck["model"])
            self.trainer.optimizer.load_state_dict(ck["optimizer"])
            self.trainer.aux_optimizer.load_state_dict(ck["aux_optimizer"])
            self.epoch_start = ck["epoch"] + 1

    def run(self):
        s = self.settings
        best = -1.0
        for epoch in range(self.epoch_start, s.n_epochs):
            if not s.val_only:
                self.trainer.run(epoch, "Train")
            if s.val_only or (epoch + 1) % s.val_frequency == 0:
                res = self.trainer.run(epoch, "Validation")
                if self.recorder is not None:
                    self.recorder.logger.info("epoch {} validation: {}".format(epoch + 1, {k: v for k, v in res.items()
                                                                                          if k != "class_IOU"}))
                    if res["IOU"] > best:
                        best = res["IOU"]
                        torch.save(self.model.state_dict(), os.path.join(self.recorder.checkpoint_path, "best_IOU_model.pth"))
            if self.recorder is not None:
                torch.save({"model": self.model.state_dict(), "optimizer": self.trainer.optimizer.state_dict(),
                            "aux_optimizer": self.trainer.aux_optimizer.state_dict(), "epoch": epoch},
                           os.path.join(self.recorder.checkpoint_path, "checkpoint.pth"))
            if s.val_only or s.is_debug:
                break


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="PMF training on MI355X")
    ap.add_argument("config_path", type=str, metavar="config_path")
    ap.add_argument("--id", type=int, default=0)
    a = ap.parse_args()
    Experiment(Option(a.config_path)).run()
