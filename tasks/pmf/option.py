"""Experiment options (tasks/pmf/option.py:10-81 of the reference): yaml keys -> attributes, raw dict kept as .config."""
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
        self.print_frequency, self.n_threads, self.experiment_id = c["print_frequency"], c["n_threads"], c["experiment_id"]
        self.dataset, self.nclasses, self.data_root, self.has_label = c["dataset"], c["nclasses"], c["data_root"], c["has_label"]
        self.n_epochs, self.batch_size, self.lr = c["n_epochs"], c["batch_size"], c["lr"]
        self.warmup_epochs, self.momentum, self.weight_decay = c["warmup_epochs"], c["momentum"], c["weight_decay"]
        self.val_only, self.is_debug, self.val_frequency = c["val_only"], c["is_debug"], c["val_frequency"]
        self.lambda_, self.gamma, self.tau = c["lambda"], c["gamma"], c["tau"]
        self.img_backbone, self.base_channels = c["img_backbone"], c["base_channels"]
        self.imagenet_pretrained = c["imagenet_pretrained"]
        self.checkpoint, self.pretrained_model = c["checkpoint"], c["pretrained_model"]
        self.save_path = os.path.join(self.save_path, "log_{}_PMFNet-{}_bs{}-lr{}_{}".format(
            self.dataset, self.img_backbone, self.batch_size[0] * self.n_gpus, self.lr, self.experiment_id))

    def check_path(self):
        os.makedirs(self.save_path, exist_ok=True)
