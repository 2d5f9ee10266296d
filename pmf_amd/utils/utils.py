"""Process-group bring-up (pc_processor/utils/utils.py:7-44): one process per GPU, RCCL over xGMI.

``backend="nccl"`` IS RCCL on PyTorch-ROCm; rendezvous is env:// (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args, backend=None):
    """fills args.rank / world_size / gpu / distributed like the reference; no-op without launcher env."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif hasattr(args, "rank") and "MASTER_ADDR" in os.environ:
        args.gpu = getattr(args, "gpu", 0) if isinstance(getattr(args, "gpu", 0), int) else 0
    else:
        args.distributed = False
        args.gpu = 0
        return
    args.distributed = True
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver
    backend = backend or getattr(args, "dist_backend", "nccl")
    if backend == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()
