"""Experiment options of tasks/epmf (the reference's tasks/epmf/option.py:10-66): the tasks/pmf keys plus ``net_type``,
``use_mtloss``, ``cls_freq`` and the ``PVconfig`` block; raw dict kept as .config."""
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
        self.net_type, self.use_mtloss = c["net_type"], c["use_mtloss"]
        self.lambda_, self.gamma, self.tau = c["lambda"], c["gamma"], c["tau"]
        self.img_backbone, self.base_channels = c["img_backbone"], c["base_channels"]
        self.imagenet_pretrained = c["imagenet_pretrained"]
        self.cls_freq = c["cls_freq"]
        self.checkpoint, self.pretrained_model = c["checkpoint"], c["pretrained_model"]
        bs = self.batch_size[0] * self.n_gpus if ("RANK" in os.environ and "WORLD_SIZE" in os.environ) else self.batch_size[0]
        self.save_path = os.path.join(self.save_path, "log_{}_{}-{}_E{}-bs{}-lr{}_{}".format(
            self.dataset, self.net_type, self.img_backbone, self.n_epochs, bs, self.lr, self.experiment_id))

    def check_path(self):
        os.makedirs(self.save_path, exist_ok=True)
