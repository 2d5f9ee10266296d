"""``pc_processor.dataset.nuScenes`` (pc_processor/dataset/nuScenes/dataset_nuscenes.py:74-282): the reader sits on the
third-party nuscenes-devkit (NuScenes tables, LidarPointCloud, view_points), which is not part of this image.  The name
resolves so that tasks/pmf/trainer.py:127-136 reaches a clear message instead of an AttributeError; everything behind the
dataset -- the per-camera-view loader (nus_perspective_loader.py), the six-camera inference loop
(tasks/pmf_eval_nuscenes/infer.py), the merge with the LiDAR-only fallback (postproc/merge.py) -- is built and tested on a
devkit-free stand-in with the same attributes (oracle/cases.py SyntheticNus, tests only)."""
from .nus_perspective_loader import NusPerspectiveViewLoader  # noqa: F401


class Nuscenes(object):
    def __init__(self, root, version="v1.0-trainval", split="train", **kw):
        try:
            import nuscenes  # noqa: F401
        except ImportError as e:
            raise ImportError("pc_processor.dataset.nuScenes.Nuscenes needs the nuscenes-devkit package (pip install "
                              "nuscenes-devkit), which is not installed in this environment") from e
        raise NotImplementedError("nuScenes table reader: install nuscenes-devkit and plug a dataset object with "
                                  "loadDataByIndex / loadImage / parsePathInfoByIndex / proj_matrix / class_map_lut "
                                  "into PerspectiveViewLoader (INTEGRATION.md)")
