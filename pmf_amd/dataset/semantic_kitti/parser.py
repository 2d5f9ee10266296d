"""SemanticKITTI on-disk formats -- call surface of pc_processor/dataset/semantic_kitti/parser.py:7-227 (host I/O only).

    <root>/<seq>/velodyne/*.bin   float32 x, y, z, intensity per point
    <root>/<seq>/labels/*.label   uint32 per point: semantic id in the low 16 bits, instance id in the high 16
    <root>/<seq>/image_2/*.png    left colour camera
    <root>/<seq>/calib.txt        "P2: 12 floats", "Tr: 12 floats" (+ others); projection = P2 (3x4) . Tr (4x4)
    predictions: <out>/sequences/<seq>/predictions/<frame>.label, int32 ORIGINAL label ids (learning_map_inv)

The LiDAR -> camera mapping (mapLidar2Camera / mapLidar2CameraCropYaw in the reference) is not a dataset method here:
it runs inside the loaders as HIP kernels (dataset/perspective_view_loader*.py) from ``proj_matrix[seq]``."""
import os

import numpy as np
import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "semantic-kitti.yaml")


def label_tables(cfg):
    """the dictionaries the reference parser reads (parser.py:100-133) from either layout of the label file: the
    semantic-kitti-api one (labels / color_map / content / learning_map / learning_map_inv / learning_ignore /
    mapped_class_name / color_map_inv) or this package's two tables (raw / train rows)."""
    if "raw" not in cfg:
        return cfg
    out = {"name": cfg.get("name", ""), "split": cfg.get("split", {})}
    out["labels"] = {r["id"]: r["name"] for r in cfg["raw"]}
    out["color_map"] = {r["id"]: list(r["color"]) for r in cfg["raw"]}
    out["content"] = {r["id"]: r["content"] for r in cfg["raw"]}
    out["learning_map"] = {r["id"]: r["train"] for r in cfg["raw"]}
    out["learning_map_inv"] = {r["id"]: r["raw"] for r in cfg["train"]}
    out["learning_ignore"] = {r["id"]: bool(r["ignore"]) for r in cfg["train"]}
    out["mapped_class_name"] = {r["id"]: r["name"] for r in cfg["train"]}
    out["color_map_inv"] = {r["id"]: list(r["color"]) for r in cfg["train"]}
    return out


class SemanticKitti(object):
    def __init__(self, root, sequences, config_path, has_image=True, has_pcd=True, has_label=True):
        self.root, self.sequences = root, sequences
        self.sequences.sort()
        self.has_label, self.has_image, self.has_pcd = has_label, has_image, has_pcd
        if not os.path.isfile(config_path):
            raise ValueError("config file not found: {}".format(config_path))
        with open(config_path, "r") as f:
            self.data_config = label_tables(yaml.safe_load(f))
        if not os.path.isdir(self.root):
            raise ValueError("dataset not found: {}".format(self.root))
        self.pointcloud_files, self.label_files, self.image_files = [], [], []
        self.proj_matrix = {}
        self.fov_left, self.fov_right = -45 / 180.0 * np.pi, 45 / 180.0 * np.pi

        def listing(seq, sub, ext):
            d = os.path.join(self.root, seq, sub)
            return [os.path.join(d, f) for f in os.listdir(d) if ext in f]
        for seq in self.sequences:
            seq = "{0:02d}".format(int(seq))
            pcs = listing(seq, "velodyne", ".bin")
            if self.has_label:
                labels = listing(seq, "labels", ".label")
                if self.has_pcd:
                    assert len(pcs) == len(labels)
                self.label_files.extend(labels)
            if self.has_image:
                images = listing(seq, "image_2", ".png")
                if self.has_pcd:
                    assert len(pcs) == len(images)
                self.image_files.extend(images)
                calib = self.read_calib(os.path.join(self.root, seq, "calib.txt"))
                self.proj_matrix[seq] = np.matmul(calib["P2"], calib["Tr"])
            self.pointcloud_files.extend(pcs)
        self.pointcloud_files.sort()
        self.label_files.sort()
        self.image_files.sort()

        cfg = self.data_config

        def colour_lut(table):
            lut = np.zeros((max(table) + 1 + 100, 3), dtype=np.float32)
            for k, v in table.items():
                lut[k] = np.array(v, np.float32) / 255.0
            return lut
        self.sem_color_lut = colour_lut(cfg["color_map"])
        self.sem_color_lut_inv = colour_lut(cfg["color_map_inv"])
        self.inst_color_map = np.random.uniform(low=0.0, high=1.0, size=(10000, 3))

        def id_lut(table):                       # +100: room for unknown label ids (parser.py:117-118)
            lut = np.zeros((max(max(table), 0) + 100), dtype=np.int32)
            for k, v in table.items():
                lut[k] = v
            return lut
        self.class_map_lut = id_lut(cfg["learning_map"])
        self.class_map_lut_inv = id_lut(cfg["learning_map_inv"])
        content = np.zeros(len(cfg["learning_map_inv"]), dtype=np.float32)
        for cl, freq in cfg["content"].items():
            content[self.class_map_lut[cl]] += freq
        self.cls_freq = content
        self.mapped_cls_name = cfg["mapped_class_name"]

    @staticmethod
    def read_calib(calib_path):
        """-> {"P2": float64 [3,4], "Tr": float64 [4,4] (last row 0 0 0 1)}; parsing stops at the first empty line"""
        rows = {}
        with open(calib_path, "r") as f:
            for line in f.readlines():
                if line == "\n":
                    break
                key, value = line.split(":", 1)
                rows[key] = np.array([float(x) for x in value.split()])
        tr = np.identity(4)
        tr[:3, :4] = rows["Tr"].reshape(3, 4)
        return {"P2": rows["P2"].reshape(3, 4), "Tr": tr}

    @staticmethod
    def readPCD(path):
        return np.fromfile(path, dtype=np.float32).reshape(-1, 4)

    @staticmethod
    def readLabel(path):
        label = np.fromfile(path, dtype=np.int32)
        return label & 0xFFFF, label >> 16

    def parsePathInfoByIndex(self, index):
        path = self.pointcloud_files[index]
        parts = path.split("\\") if "\\" in path else path.split("/")
        return parts[-3], parts[-1].split(".")[0]

    def labelMapping(self, label):
        return self.class_map_lut[label]

    def loadLabelByIndex(self, index):
        return self.readLabel(self.label_files[index])

    def loadDataByIndex(self, index):
        pointcloud = self.readPCD(self.pointcloud_files[index])
        if self.has_label:
            sem_label, inst_label = self.readLabel(self.label_files[index])
        else:
            sem_label = np.zeros(pointcloud.shape[0], dtype=np.int32)
            inst_label = np.zeros(pointcloud.shape[0], dtype=np.int32)
        return pointcloud, sem_label, inst_label

    def loadImage(self, index):
        from PIL import Image
        return Image.open(self.image_files[index])

    def __len__(self):
        return len(self.pointcloud_files)


def write_prediction(dataset, index, pred, prediction_path):
    """tasks/pmf_eval_semantickitti/infer.py:126-146: learning ids -> original ids (learning_map_inv), int32, one file
    per frame under <prediction_path>/sequences/<seq>/predictions/<frame>.label; returns the path."""
    pred = np.asarray(pred).reshape(-1).astype(np.int32)
    seq_id, frame_id = dataset.parsePathInfoByIndex(index)
    out_dir = os.path.join(prediction_path, "sequences", seq_id, "predictions")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "{}.label".format(frame_id))
    dataset.class_map_lut_inv[pred].tofile(path)
    return path
