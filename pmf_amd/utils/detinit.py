"""Closed-form deterministic weights / inputs.

Tests, smoke() and bench.py need identical parameters in three places (the
reference modules when fixtures are generated, the CPU oracle, and the HIP
model on the GPU box) without shipping a 146 MB checkpoint.  Values are a pure
function of (state-dict key, flat element index): an integer hash mapped to a
uniform, scaled so activations stay O(1) through ~110 conv layers.  No torch RNG
is involved, so the GPU box regenerates them bit-identically.
"""
import zlib

import numpy as np
import torch


def _hash_uniform(key: str, n: int, salt: int = 0) -> np.ndarray:
    """n float64 uniforms in [0, 1) from a 32-bit mix of (crc32(key), index)."""
    seed = (zlib.crc32(key.encode()) + 0x9E3779B9 * (salt + 1)) & 0xFFFFFFFF
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.float64) / 4294967296.0


def det_tensor(key: str, shape, lo=-1.0, hi=1.0, salt=0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _hash_uniform(key, n, salt)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


@torch.no_grad()
def deterministic_init(model: torch.nn.Module, salt: int = 0) -> torch.nn.Module:
    """Overwrite every parameter and buffer of ``model`` in place (state-dict key order irrelevant)."""
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            v.zero_()
        elif k.endswith("running_mean"):
            v.copy_(det_tensor(k, v.shape, -0.2, 0.2, salt))
        elif k.endswith("running_var"):
            v.copy_(det_tensor(k, v.shape, 0.6, 1.4, salt))
        elif v.dim() == 4:                       # conv weight: variance-preserving uniform
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            b = float(np.sqrt(3.0 / fan_in)) * 1.2
            v.copy_(det_tensor(k, v.shape, -b, b, salt))
        elif v.dim() == 1 and _is_bn_key(k, sd):
            if k.endswith("weight"):
                v.copy_(det_tensor(k, v.shape, 0.6, 1.4, salt))
            else:
                v.copy_(det_tensor(k, v.shape, -0.2, 0.2, salt))
        else:                                    # conv bias
            v.copy_(det_tensor(k, v.shape, -0.1, 0.1, salt))
    return model


def _is_bn_key(k: str, sd) -> bool:
    """A 1-D weight/bias belongs to a BatchNorm iff a sibling running_mean exists."""
    return (k.rsplit(".", 1)[0] + ".running_mean") in sd


def synthetic_batch(n, h, w, nclasses=20, seed=1, fill=0.15, device="cpu"):
    """SURVEY 8(d) config inputs: pcd = N(0,1)*mask, rgb = U[0,1), label = randint*mask.

    Generated with a numpy PCG64 stream (bit-reproducible across machines)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mask = (rng.random((n, 1, h, w)) < fill).astype(np.float32)
    pcd = rng.standard_normal((n, 5, h, w)).astype(np.float32) * mask
    rgb = rng.random((n, 3, h, w)).astype(np.float32)
    label = (rng.integers(0, nclasses, (n, h, w)) * mask[:, 0]).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    return t(pcd), t(rgb), t(label), t(mask[:, 0])
