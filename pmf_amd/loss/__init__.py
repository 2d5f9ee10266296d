from .focal_softmax import FocalSoftmaxLoss  # noqa: F401
from .lovasz_softmax import Lovasz_softmax, lovasz_softmax  # noqa: F401
from .perception import (perception_aware_loss, normalized_entropy, pmf_total_loss, epmf_total_loss,  # noqa: F401
                         EPMF_TERMS)
from .fused import pmf_total_loss_fused, weighted_loss_fused  # noqa: F401
from .multi_task_loss import MultiTaskLoss  # noqa: F401
