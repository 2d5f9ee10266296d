"""``pc_processor.dataset.nuScenes`` (pc_processor/dataset/nuScenes/dataset_nuscenes.py:74-282): the reader sits on the
third-party nuscenes-devkit (NuScenes tables, LidarPointCloud, view_points), which is not part of this image.  The name
resolves so that tasks/pmf/trainer.py:127-136 reaches a clear message instead of an AttributeError; everything behind the
dataset -- projection, scatter, the 6-camera merge with the LiDAR-only fallback -- is built and tested
(dataset/perspective_view_loader.py, postproc/merge.py)."""


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
