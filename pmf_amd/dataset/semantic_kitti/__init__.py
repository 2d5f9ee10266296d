from .parser import SemanticKitti, write_prediction  # noqa: F401
