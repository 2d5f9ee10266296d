"""``from nus_perspective_loader import NusPerspectiveViewLoader`` as tasks/pmf_eval_nuscenes/infer.py:14 of the reference
has it: the class lives in the package (pmf_amd/dataset/nuScenes/nus_perspective_loader.py)."""
from pmf_amd.dataset.nuScenes import NusPerspectiveViewLoader  # noqa: F401
