"""Building blocks of a plan (pmf_amd/plan.py): arenas and buffers, materialised tensors (T), per-pixel masks (PM) and views
(V: a tensor seen through BatchNorm-apply / ReLU / a Dropout2d multiplier, folded into the consumers' loads)."""
import torch

_A = 256  # arena alignment (bytes)
SPLITK_BYTES = 32 << 20      # shared split-K scratch
DBIAS_LD = 2048              # floats per partial conv-bias-gradient row (max Cout)
COL_ROWS = 512               # PMF_COL_ROWS in csrc/common.h
RED_BATCH = 32               # weight-gradient reductions per batched stage-2 launch (flat training state);
                             # measured 8: 24.80, 16: 24.77, 32: 24.70, one launch: 24.65 ms per step -- 32 keeps four
                             # launches per backward pass so data-parallel ranges still become final early


def _ru(a, b):
    return (a + b - 1) // b * b


class Buf:
    __slots__ = ("arena", "off", "nbytes")

    def __init__(self, arena, off, nbytes):
        self.arena, self.off, self.nbytes = arena, off, nbytes

    @property
    def ptr(self):
        return self.arena.base + self.off

    def at(self, float_off):
        return self.ptr + 4 * float_off

    def tensor(self, shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= s
        esz = torch.empty(0, dtype=dtype).element_size()
        return self.arena.t[self.off:self.off + n * esz].view(dtype).view(shape)




class ExternalBuf:
    """a caller-owned device tensor seen through the Buf interface (the flat gradient buffer of FlatState)."""

    def __init__(self, t):
        self.t = t
        self.nbytes = t.numel() * t.element_size()

    @property
    def ptr(self):
        return self.t.data_ptr()

    def at(self, float_off):
        return self.ptr + 4 * float_off

    def tensor(self, shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= s
        return self.t.view(-1)[:n].view(shape)


class Arena:
    def __init__(self, name):
        self.name, self.size, self.t, self.base = name, 0, None, 0

    def alloc(self, nbytes):
        off = self.size
        self.size = _ru(off + max(int(nbytes), 4), _A)
        return Buf(self, off, int(nbytes))

    def materialise(self, device, zero=False):
        n = max(self.size, _A)
        self.t = (torch.zeros if zero else torch.empty)(n, dtype=torch.uint8, device=device)
        self.base = self.t.data_ptr()


class T:
    """Materialised NHWC tensor [N,H,W,ldc] with C logical channels (ldc = C rounded up to 8)."""

    def __init__(self, plan, N, H, W, C, name="", arena=None, ldc=None):
        self.N, self.H, self.W, self.C = N, H, W, C
        self.ldc = ldc or _ru(C, 8)
        self.name = name
        self.buf = (arena or plan.act).alloc(4 * N * H * W * self.ldc)
        if name:
            plan.tensors[name] = self
        self.g = None            # gradient tensor (same geometry)
        self.g_written = False
        self.needs_grad = True
        self.lane = plan.lane    # lane of the op that produces it (its backward runs there too)

    @property
    def npix(self):
        return self.N * self.H * self.W


class PM:
    """per-pixel validity mask, dense float [N, H, W] (EPMF SparseVariantConv, epmf_net.py:30-50)."""

    def __init__(self, plan, N, H, W):
        self.N, self.H, self.W = N, H, W
        self.buf = plan.act.alloc(4 * N * H * W)


class V:
    def __init__(self, t, scale=None, shift=None, cmul=None, cmul_ld=0, relu=False, bcast=False, bn=None):
        self.t, self.scale, self.shift = t, scale, shift
        self.cmul, self.cmul_ld = cmul, cmul_ld     # cmul: (Buf, float offset) or None
        self.relu, self.bcast, self.bn = relu, bcast, bn
        self.gy = None
        self.gy_written = False

    def with_cmul(self, cm, ld):
        v = V(self.t, self.scale, self.shift, cm, ld, self.relu, self.bcast, self.bn)
        v._parent = self
        return v

    def root(self):
        return getattr(self, "_parent", self)
