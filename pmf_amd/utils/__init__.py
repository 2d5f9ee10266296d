from .utils import init_distributed_mode, is_main_process, get_rank, get_world_size, is_dist_avail_and_initialized  # noqa: F401
from .warmup_lr import WarmupCosineLR  # noqa: F401
from .avgmeter import AverageMeter, RemainTime  # noqa: F401
from .detinit import deterministic_init, det_tensor, synthetic_batch  # noqa: F401
