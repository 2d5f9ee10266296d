from .pmf_net import PMFNet, SalsaNext, SalsaNextFusion, ResNet, RGBDecoder, ASPP, ResidualBasedFusionBlock  # noqa: F401
from .epmf_net import EPMFNet  # noqa: F401
