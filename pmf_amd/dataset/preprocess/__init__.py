from . import augmentor, projection  # noqa: F401
