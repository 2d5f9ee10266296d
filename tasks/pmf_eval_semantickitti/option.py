"""Options of the inference task (tasks/pmf_eval_semantickitti/option.py of the reference): yaml keys -> attributes."""
import os

import yaml


class Option(object):
    def __init__(self, config_path):
        self.config_path = config_path
        with open(config_path, "r") as f:
            self.config = yaml.safe_load(f)
        c = self.config
        self.save_path, self.seed, self.gpu = c["save_path"], c["seed"], str(c["gpu"])
        self.n_threads, self.is_debug = c["n_threads"], c["is_debug"]
        self.dataset, self.n_classes, self.nclasses = c["dataset"], c["nclasses"], c["nclasses"]
        self.data_root, self.has_label = c["data_root"], c["has_label"]
        self.img_backbone, self.base_channels = c["img_backbone"], c["base_channels"]
        self.imagenet_pretrained = c.get("imagenet_pretrained", False)
        self.pretrained_model = c["pretrained_model"]
        self.save_path = os.path.join(self.save_path, "Eval_{}_PMFNet-{}_{}".format(
            self.dataset, self.img_backbone, c.get("experiment_id", "0")))

    def check_path(self):
        os.makedirs(self.save_path, exist_ok=True)
