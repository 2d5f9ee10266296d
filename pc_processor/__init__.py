"""`import pc_processor` compatibility surface -- every name resolves to pmf_amd."""
from pmf_amd import models, postproc, dataset, loss, metrics, utils, layers, checkpoint  # noqa: F401
import sys as _sys
for _n in ("models", "postproc", "dataset", "loss", "metrics", "utils", "layers", "checkpoint"):
    _sys.modules["pc_processor." + _n] = globals()[_n]
_sys.modules["pc_processor.layers.sync_bn"] = layers.sync_bn
