"""Recorder (pc_processor/checkpoint/recorder.py:7-84): log directory, python logger, optional tensorboard."""
import logging
import os
import sys


class _NullBoard:
    def add_scalar(self, *a, **k):
        pass

    add_image = add_scalar

    def close(self):
        pass


class Recorder:
    def __init__(self, settings, save_path, use_tensorboard=True):
        self.settings, self.save_path = settings, save_path
        self.checkpoint_path = os.path.join(save_path, "checkpoint")
        self.log_path = os.path.join(save_path, "log")
        for p in (self.save_path, self.checkpoint_path, self.log_path):
            os.makedirs(p, exist_ok=True)
        self.logger = logging.getLogger("pmf_amd.%s" % os.path.basename(os.path.normpath(save_path)))
        self.logger.setLevel(logging.INFO)
        if not self.logger.handlers:
            fmt = logging.Formatter("%(asctime)s %(message)s")
            for h in (logging.StreamHandler(sys.stdout), logging.FileHandler(os.path.join(self.log_path, "console.log"))):
                h.setFormatter(fmt)
                self.logger.addHandler(h)
        self.tensorboard = _NullBoard()
        if use_tensorboard:
            try:
                from tensorboardX import SummaryWriter   # optional, absent in this image
                self.tensorboard = SummaryWriter(self.log_path)
            except Exception:
                self.logger.info("tensorboardX unavailable: scalar logging goes to console.log only")
        with open(os.path.join(self.log_path, "settings.log"), "w") as f:
            for k, v in sorted(vars(settings).items()):
                if k != "config":
                    f.write("%s: %s\n" % (k, v))
