import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The plan autotuner picks tile configurations from launch timings, i.e. not reproducibly across processes; the parity
# tests run the built-in heuristics (reproducible), tests/test_gpu_ops.py::test_conv_tile_configs pins EVERY
# configuration the tuner can choose, and test_gpu_parity.py::test_autotuned_plan_matches_heuristic_plan runs it.
os.environ.setdefault("PMF_AUTOTUNE", "0")


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
