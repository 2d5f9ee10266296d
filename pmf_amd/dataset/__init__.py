from .perspective_view_loader import PerspectiveViewLoader, project_frame_gpu, center_crop_pad_gpu  # noqa: F401
