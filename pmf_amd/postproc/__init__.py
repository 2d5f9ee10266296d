from .knn import KNN  # noqa: F401
