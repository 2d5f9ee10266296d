from .perspective_view_loader import (PerspectiveViewLoader, FlipRotateCrop, ColorJitter, project_frame_gpu,  # noqa: F401
                                      center_crop_pad_gpu)
from .perspective_view_loader_v2 import PerspectiveViewLoaderV2, project_frame_v2_gpu  # noqa: F401
from .salsanext_loader import SalsaNextLoader  # noqa: F401
from .preprocess import augmentor, projection  # noqa: F401
from . import semantic_kitti  # noqa: F401
from . import nuScenes  # noqa: F401
