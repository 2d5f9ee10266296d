"""pmf_amd -- MI355X-native implementation of the PMF dual-branch fusion hot path."""
from . import models, postproc, dataset, loss, metrics, utils, layers, checkpoint  # noqa: F401
