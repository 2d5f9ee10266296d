"""`import pc_processor` compatibility surface -- every name resolves to pmf_amd."""
from pmf_amd import models, postproc, dataset, loss, metrics, utils, layers, checkpoint  # noqa: F401
import sys as _sys
for _n in ("models", "postproc", "dataset", "loss", "metrics", "utils", "layers", "checkpoint"):
    _sys.modules["pc_processor." + _n] = globals()[_n]
_sys.modules["pc_processor.layers.sync_bn"] = layers.sync_bn
for _n in ("salsanext_loader", "perspective_view_loader", "perspective_view_loader_v2", "semantic_kitti", "preprocess",
           "nuScenes"):
    _sys.modules["pc_processor.dataset." + _n] = getattr(dataset, _n)
_sys.modules["pc_processor.dataset.semantic_kitti.parser"] = dataset.semantic_kitti.parser
_sys.modules["pc_processor.dataset.preprocess.augmentor"] = dataset.preprocess.augmentor
_sys.modules["pc_processor.dataset.preprocess.projection"] = dataset.preprocess.projection
