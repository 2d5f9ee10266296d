"""Experiment options of tasks/salsanext (the reference's tasks/salsanext/option.py:10-58): note ``n_classes`` (not
``nclasses``) and a scalar ``batch_size`` in this task; raw dict kept as .config."""
import os

import yaml


class Option(object):
    def __init__(self, config_path):
        self.config_path = config_path
        with open(config_path, "r") as f:
            self.config = yaml.safe_load(f)
        c = self.config
        self.save_path, self.seed, self.gpu = c["save_path"], c["seed"], str(c["gpu"])
        self.rank, self.world_size, self.distributed = 0, 1, False
        self.n_gpus = len(self.gpu.split(","))
        self.dist_backend, self.dist_url = "nccl", "env://"       # "nccl" is RCCL on ROCm
        self.print_frequency, self.n_threads, self.experiment_id = int(c["print_frequency"]), c["n_threads"], c["experiment_id"]
        self.dataset, self.n_classes, self.data_root, self.has_label = c["dataset"], c["n_classes"], c["data_root"], c["has_label"]
        self.n_epochs, self.batch_size, self.lr = c["n_epochs"], c["batch_size"], c["lr"]
        self.warmup_epochs, self.momentum, self.weight_decay = c["warmup_epochs"], c["momentum"], c["weight_decay"]
        self.val_only, self.is_debug, self.val_frequency = c["val_only"], c["is_debug"], c["val_frequency"]
        self.net_type = c["net_type"]
        self.checkpoint, self.pretrained_model = c["checkpoint"], c["pretrained_model"]
        self.save_path = os.path.join(self.save_path, "log_{}_{}_bs{}_ep{}_lr{}_{}".format(
            self.dataset, self.net_type, self.batch_size * self.n_gpus, self.n_epochs, self.lr, self.experiment_id))

    def check_path(self):
        os.makedirs(self.save_path, exist_ok=True)
