from .recorder import Recorder  # noqa: F401
