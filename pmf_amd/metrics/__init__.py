from .iou_eval import IOUEval  # noqa: F401
