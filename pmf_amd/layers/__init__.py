from . import sync_bn  # noqa: F401
