import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Tile configurations: the tests run the SHIPPED table (pmf_amd/tuned/gfx950.txt: the tuner's choices for every BASELINE shape,
# what bench.py and a user run) and the built-in heuristics for shapes the table does not know -- reproducible across processes,
# no timing inside the tests.  tests/test_gpu_ops.py::test_conv_tile_configs pins EVERY configuration the tuner can choose,
# test_gpu_fullsize.py runs the full-size plans with the heuristics ("0") and with the shipped table ("cache").
os.environ.setdefault("PMF_AUTOTUNE", "cache")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _G:
        def __init__(self):
            self._c = {}

        def __call__(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
            return self._c[name]
    return _G()
