"""Options of the nuScenes inference task (tasks/pmf_eval_nuscenes/option.py:10-64 of the reference): yaml keys -> attributes;
results go to <pretrained_path>/Eval-<dataset>-PMFNet-<model>-<KNN-n|noKNN>-<experiment_id>."""
import os
import shutil

import yaml


class Option(object):
    def __init__(self, config_path):
        self.config_path = config_path
        with open(config_path, "r") as f:
            self.config = yaml.safe_load(f)
        c = self.config
        self.save_path = c["pretrained_path"]
        self.seed, self.gpu = c["seed"], str(c["gpu"])
        self.rank, self.world_size, self.distributed = 0, 1, False
        self.print_frequency, self.n_threads = c["print_frequency"], c["n_threads"]
        self.experiment_id, self.is_debug = c["experiment_id"], c["is_debug"]
        self.dataset, self.n_classes, self.nclasses = c["dataset"], c["nclasses"], c["nclasses"]
        self.data_root, self.has_label = c["data_root"], c["has_label"]
        self.base_channels, self.img_backbone = c["base_channels"], c["img_backbone"]
        self.imagenet_pretrained = c["imagenet_pretrained"]
        self.pretrained_model = os.path.join(c["pretrained_path"], "checkpoint", c["best_model"])
        if not os.path.isdir(self.save_path):
            raise ValueError("pretrained model is required, please train your model first. Path not exist: {}".format(
                self.save_path))
        knn = c["post"]["KNN"]
        knn_str = "KNN-{}".format(knn["params"]["search"]) if knn["use"] else "noKNN"
        self.save_path = os.path.join(self.save_path, "Eval-{}-PMFNet-{}-{}-{}".format(
            self.dataset, c["best_model"].strip(".pth"), knn_str, self.experiment_id))

    def check_path(self, overwrite=False):
        if os.path.exists(self.save_path):
            if not overwrite:
                raise OSError("Directory exits: {}".format(self.save_path))
            shutil.rmtree(self.save_path)
        os.makedirs(self.save_path)
