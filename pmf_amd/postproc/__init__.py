from .knn import KNN  # noqa: F401
from .merge import getMergePred  # noqa: F401
