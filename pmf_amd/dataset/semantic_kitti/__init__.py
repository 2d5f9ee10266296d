from .parser import SemanticKitti, write_prediction, DEFAULT_CONFIG  # noqa: F401
