"""Static execution plans for the PMF network on MI355X.

A plan is built once per (model, N, H, W, training) and holds
  * arenas (one device allocation each) for activations, gradients, packed weights and BN scratch,
  * a flat ctypes array of ``pmf_op_t`` for the forward pass and one for the backward pass.
Running a pass is ONE call into libpmf_amd.so (``pmf_plan_run``) that enqueues every kernel on the
current HIP stream.  There is no tracing compiler and no per-op Python at run time.

Graph conventions
  T  -- a materialised NHWC fp32 tensor.
  V  -- a *view*: T seen through BatchNorm-apply (scale/shift), optional ReLU and an optional
        Dropout2d (n,c) multiplier.  Consumers fold the view into their loads; nothing is materialised.
  For a view with BatchNorm, gradients are delivered w.r.t. the BN OUTPUT y (``V.gy``); the BN entry on the
  tape turns them into the gradient w.r.t. the conv pre-activation (``T.g``), which feeds dgrad / wgrad.
"""
import ctypes as C
import os

import torch

from . import _lib as L

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


_TUNED = {}      # process-wide autotuner cache: conv shape key -> tile configuration (Plan.autotune)


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


class Plan:
    def __init__(self, device, training, flat=None, dry=False):
        self.device, self.training = device, training
        self.flat = flat                    # FlatState: parameter gradients go to its buffer at its offsets
        self.dry = dry                      # dry run: walk the tape (records the backward parameter order), no memory
        self.act = Arena("act")
        self.zero_fwd = Arena("zero_fwd")
        self.zero_bwd = Arena("zero_bwd")
        self.persist = Arena("persist")     # packed weights (padding zeroed once), BN scale/shift/saved stats
        self.fwd, self.bwd = [], []         # deferred op builders: callables returning L.Op
        self.tape = []
        self.pack_jobs = []
        self.wg_scratch = 0                 # bytes of the shared wgrad partial-sum workspace
        self.params = []                    # (param, grad Buf float offset)
        self._graphs, self._graph_seen = {}, {}   # hipGraph replay cache (see run)
        # fp32 convolutions on the bf16 matrix pipe (three-way operand split, six products; PMF_CONV_F32=1: fp32 MFMA only)
        self.s3 = os.environ.get("PMF_CONV_F32", "0") != "1"
        # 1x1 layers never use the LDS-staged split kernel (the split + store of the input tile costs more than the 2.67x
        # shorter MFMA phase saves: 103 vs 149 us on the 192 -> 64 concat conv); they have their own variant that reads
        # the activations straight from global memory (s3_direct_min_pix, 67 us on that layer)
        self.s3_min_taps = int(os.environ.get("PMF_S3_MIN_TAPS", "2"))
        self.s3_direct_min_pix = int(os.environ.get("PMF_S3_DIRECT_MIN_PIX", "1"))   # 0: 1x1 layers stay on fp32 MFMA
        self._conv_fold = {}                # backward conv op index -> index of the BN-backward fold op reading its rows
        self.bn_bwd_fused = os.environ.get("PMF_BN_BWD_FUSED", "1") != "0"
        self._conv_fin = {}                 # forward conv op index -> index of the BN finalize op reading its rows
        self.n_wgrad = 0                    # weight-gradient ops emitted so far
        import os as _os
        # lanes: independent branches of the network on separate HIP streams (csrc/plan.cpp); ``lane`` is the lane ops
        # are being emitted on; cross-lane edges are plan events (record after one op, wait before another)
        self.lane = 0
        self.n_lanes = 4
        # weight gradients are leaves of the backward graph (only the optimiser / the gradient all-reduce reads them):
        # they run on lanes of their own (lane 0 -> 2, lane 1 -> 3; PMF_WGRAD_LANE=2: one shared lane, 0: inline on
        # the home lane), in batches of PMF_WGRAD_BATCH behind ONE event of the home lane, so that the input-gradient
        # chain -- the critical path -- is not queued behind them and their machine-filling launches run under its
        # latency-bound ones.  (Round 2 measured this slower and kept it off: what it measured was the multi-branch
        # hipGraph replay, see csrc/plan.cpp; with the range replayed as linear pieces on real streams it is worth
        # 1.0 ms of 17.7 ms per step.)
        self.wgrad_lane = int(_os.environ.get("PMF_WGRAD_LANE", "23"))
        self._wgrad_lane_of = (lambda home: 2 + (home & 1)) if self.wgrad_lane == 23 else (lambda home: self.wgrad_lane)
        self.wgrad_batch = int(_os.environ.get("PMF_WGRAD_BATCH", "4"))
        self._wg_deferred = {}
        # PMF_WGRAD_POLICY=phase: only the weight gradients of the FIRST full-resolution stretch of a lane's backward
        # are deferred; they are released in one batch at the lane's first small-map layer (<= PMF_WGRAD_THIN_PIX output
        # pixels), whose latency-bound launches leave most of the chip to them; everything after runs inline
        self.wgrad_policy = _os.environ.get("PMF_WGRAD_POLICY", "batch")
        self.wgrad_homes = int(_os.environ.get("PMF_WGRAD_HOMES", "3"))      # bit h: lane h defers its weight gradients
        self.wgrad_thin_pix = int(_os.environ.get("PMF_WGRAD_THIN_PIX", "16384"))
        self._wg_phase_done = {}
        # PMF_WGRAD_DELAY=n: a released batch is emitted n home-lane ops AFTER its release point (it still waits only for
        # the event at the release point).  hipGraph replay enqueues nodes in capture order and resolves a cross-stream
        # edge against the source stream's tail at that moment: a batch captured directly behind its release point makes
        # the home lane's next op wait for the batch's first launches (measured: 0.5 ms stalls of the critical path).
        self.wgrad_delay = int(_os.environ.get("PMF_WGRAD_DELAY", "0"))
        self._wg_armed = {}                 # home lane -> [closures, event, home ops still to emit first]
        self.n_events = 0
        self._event_pos = {}                # event -> list position of its record op
        self._last_op = {}                  # (id(op list), lane) -> last entry emitted on that lane
        self._pending_wait = {}             # (id(op list), lane) -> (event, position of its record op)
        self._touched = None                # gradient tensors touched by the tape entry being emitted
        self.red_batch = int(_os.environ.get("PMF_RED_BATCH", str(RED_BATCH)))     # 0: one reduction op per layer
        self.batch_reds = self.red_batch > 0
        self.pending_reds, self._red_tables = {}, []     # lane -> queued stage-2 reductions
        self.dp_events = []                 # (backward op count behind a batched reduction, [plan events]) -- see flush_reds
        self.grad_done = {}                 # id(param) -> index (in self.bwd) of the last op writing its gradient
        self.pgrad_floats = 0
        self._pid = {}
        self.masks = None                   # dropout multipliers (torch tensor), laid out by MaskLayout
        self.masks_ptr = 0
        self.bn_modules = []
        self.in_slots, self.out_slots = {}, {}
        self.tensors, self.views = {}, {}   # debug registry: name -> T / V
        self.meta_fwd, self.meta_bwd = {}, {}   # op index (before prologue shift) -> dict(family, flops)
        self.colrows_max = 4

    # ------------------------------------------------------------------ parameter bookkeeping
    def pgrad(self, p):
        """float offset of p's gradient inside the flat gradient buffer."""
        if id(p) not in self._pid:
            if self.flat is not None:
                self._pid[id(p)] = (len(self.params), self.flat.offset[id(p)])
            else:
                self._pid[id(p)] = (len(self.params), self.pgrad_floats)
                self.pgrad_floats += _ru(p.numel(), 64)
            self.params.append(p)
        return self._pid[id(p)][1]

    # ------------------------------------------------------------------ op emission helpers
    def emit(self, lst, kind, fill):
        """fill(op) populates a zeroed L.Op at finalise time (pointers are known only then).  The op runs on the
        current lane; a wait registered for this lane (wait_event) is attached to it."""
        arm = self._wg_armed.get(self.lane) if (self._wg_armed and lst is self.bwd) else None
        if arm is not None:
            if arm[2] > 0:
                arm[2] -= 1
            else:
                self._release_armed(self.lane)
        ent = [kind, fill, self.lane]       # [kind, fill, scheduling bits (pmf_amd.h)]
        key = (id(lst), self.lane)
        w = self._pending_wait.pop(key, None)
        if w is not None:
            ent[2] |= (w[0] + 1) << 8
        lst.append(ent)
        ent.append(len(lst) - 1)            # [3]: position at emission time (orders record points of one lane)
        self._last_op[key] = ent

    def note_bytes(self, lst, family, nbytes):
        """algorithmic HBM bytes (every operand read once, every result written once) of the op just
        emitted -- what bench.py divides by the measured duration for the bandwidth-bound families"""
        meta = self.meta_fwd if lst is self.fwd else self.meta_bwd
        meta[len(lst) - 1] = dict(family=family, flops=0.0, bytes=float(nbytes))

    def record_event(self, lst, lane=None):
        """plan event recorded after the last op emitted so far on ``lane`` (default: the current lane); None when the
        lane has not emitted anything into ``lst`` yet (nothing to wait for: a side lane forks from the main stream)."""
        ent = self._last_op.get((id(lst), self.lane if lane is None else lane))
        return None if ent is None else self._event_after(ent)

    def _event_after(self, ent):
        e = ((ent[2] >> 16) & 0xff) - 1
        if e < 0:
            e = self.n_events
            self.n_events += 1
            if e >= 255:
                raise RuntimeError("plan: more than 255 cross-lane events")
            ent[2] |= (e + 1) << 16
            self._event_pos[e] = ent[3]
        return (e, ent[3])

    def wait_event(self, lst, ev):
        """the NEXT op emitted on the current lane first waits for ``ev`` (from record_event).  Two lanes only: of
        several waits for the other lane the latest record point subsumes the others."""
        if ev is None:
            return
        key = (id(lst), self.lane)
        cur = self._pending_wait.get(key)
        if cur is not None and self._event_lane(lst, cur) != self._event_lane(lst, ev):
            # list position implies happens-before only along ONE lane: two pending waits on different source lanes
            # cannot be folded into one slot (an op carries a single wait)
            raise RuntimeError("plan: an op would have to wait for events of two different lanes (%r, %r)" % (cur, ev))
        if cur is None or ev[1] > cur[1]:
            self._pending_wait[key] = ev

    @staticmethod
    def _event_lane(lst, ev):
        """lane of the op behind which plan event ``ev`` = (id, list position) is recorded"""
        return lst[ev[1]][2] & 3

    def on_backward(self, fn):
        """register the backward of the op(s) just emitted; it is emitted on the lane of its forward."""
        self.tape.append((self.lane, fn))

    def _emit_tape(self):
        """walk the tape in reverse.  Gradient tensors touched by an entry (grad_of / tgrad) are tracked per entry: when
        the previous entry that touched one of them ran on the other lane, the entry's first op waits for an event
        recorded after that entry's last op -- accumulation order and read-after-write across lanes stay exactly those
        of the single-stream order."""
        lst = self.bwd
        for lane, fn in reversed(self.tape):
            self.lane = lane
            a = len(lst)
            self._touched = []
            fn()
            touched, self._touched = self._touched, None
            b = len(lst)
            if b == a:
                continue
            own = [e for e in lst[a:b] if (e[2] & 3) == lane]      # (weight gradients sit on their own lane)
            if not own:
                continue
            first, last, wait, wait_lane = own[0], own[-1], None, None
            for g in touched:
                lw = getattr(g, "_last_touch", None)
                if lw is not None and lw[0] != lane:
                    if wait_lane is not None and lw[0] != wait_lane:
                        # the latest position subsumes earlier ones only along one lane (see wait_event)
                        raise RuntimeError("plan: backward entry on lane %d depends on gradients last written on lanes %d "
                                           "and %d; an op carries one wait" % (lane, wait_lane, lw[0]))
                    wait_lane = lw[0]
                    ev = self._event_after(lw[1])
                    if wait is None or ev[1] > wait[1]:
                        wait = ev
            if wait is not None:
                cur = ((first[2] >> 8) & 0xff) - 1
                if cur < 0 or wait[1] > self._event_pos[cur]:
                    first[2] = (first[2] & ~0xff00) | ((wait[0] + 1) << 8)
            for g in touched:
                g._last_touch = (lane, last)
        for home in sorted(set(self._wg_deferred) | set(self._wg_armed)):
            self.flush_wgrads(home, final=True)
        for lane in sorted(self.pending_reds):
            self.flush_reds(lane)
        self.lane = 0

    def _touch(self, g):
        if self._touched is not None and g is not None:
            self._touched.append(g)

    def view_struct(self, v, dst):
        dst.x = v.t.buf.ptr
        dst.scale = v.scale.ptr if v.scale is not None else None
        dst.shift = v.shift.ptr if v.shift is not None else None
        if v.cmul is not None:
            dst.cmul = self.masks_ptr + 4 * v.cmul
            dst.cmul_ld = v.cmul_ld
        else:
            dst.cmul, dst.cmul_ld = None, 0
        dst.ldc = v.t.ldc
        dst.flags = (L.SRC_RELU if v.relu else 0) | (L.SRC_BCAST if v.bcast else 0)

    def src_struct(self, v, dst, C_override=None):
        self.view_struct(v, dst)
        dst.C = C_override or _ru(v.t.C, 8)
        dst.H, dst.W = v.t.H, v.t.W

    # gradient targets ------------------------------------------------------------------
    def grad_of(self, v):
        """(tensor receiving dL/dy of view v, accumulate flag); allocates lazily.
        A consumer on another lane than the producer accumulates into a PRIVATE tensor of its lane; the producer's
        backward folds it in before it reads the gradient (_fold_side).  The two lanes then only meet where the data
        dependency is (the fold), not at every accumulation into the shared tensor."""
        r = v.root()
        if self.lane != r.t.lane:
            holder = r if r.bn is not None else r.t
            side = holder.__dict__.setdefault("_side", {})
            ent = side.get(self.lane)
            if ent is None:
                ent = side[self.lane] = [T(self, r.t.N, r.t.H, r.t.W, r.t.C, r.t.name + ".gside%d" % self.lane,
                                           ldc=r.t.ldc), False]
            acc, ent[1] = ent[1], True
            self._touch(ent[0])
            return ent[0], int(acc)
        if r.bn is not None:
            if r.gy is None:
                r.gy = T(self, r.t.N, r.t.H, r.t.W, r.t.C, r.t.name + ".gy", ldc=r.t.ldc)
            acc = r.gy_written
            r.gy_written = True
            r._gy_last = None       # (_dgrad re-arms it when this writer can carry the BN-backward reduction)
            self._touch(r.gy)
            return r.gy, int(acc)
        t = r.t
        if t.g is None:
            t.g = T(self, t.N, t.H, t.W, t.C, t.name + ".g", ldc=t.ldc)
        acc = t.g_written
        t.g_written = True
        self._touch(t.g)
        return t.g, int(acc)

    def _fold_side(self, holder, g, written):
        """g (+)= the private accumulators other lanes kept for this gradient (grad_of); returns True when g holds a
        gradient afterwards.  Emitted on the current lane = the producer's lane, at the start of its backward."""
        for lane, (sg, w) in sorted(getattr(holder, "_side", {}).items()):
            if not w:
                continue
            self._touch(sg)

            def f(op, sg=sg, g=g, both=written):
                s = op.u.sm
                for k, t in enumerate((sg, g) if both else (sg,)):
                    s.v[k].x, s.v[k].ldc = t.buf.ptr, t.ldc
                s.p[0] = g.buf.ptr
                s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = L.ACT_NONE, g.ldc, g.H * g.W, int(both), _ru(g.C, 4)
                s.l[0] = g.npix
            self.emit(self.bwd, L.OP_ADD_ACT, f)
            written = True
        return written

    def tgrad(self, t):
        """the gradient tensor of a materialised T that this op is about to CONSUME."""
        if getattr(t, "_side", None):
            if t.g is None:
                t.g = T(self, t.N, t.H, t.W, t.C, t.name + ".g", ldc=t.ldc)
            t.g_written = self._fold_side(t, t.g, t.g_written)
            t._side = None
        if t.g is None or not t.g_written:
            raise RuntimeError("plan: gradient of %s consumed before any producer wrote it" % t.name)
        self._touch(t.g)
        return t.g

    # ------------------------------------------------------------------ primitives
    def fill(self, lst, buf, nfloats, value=0.0):
        def f(op):
            a = op.u.sm
            a.p[0], a.f[0], a.l[0] = buf.ptr, value, nfloats
        self.emit(lst, L.OP_FILL, f)

    def taps(self, kh, kw, dil, pad):
        out = []
        for ky in range(kh):
            for kx in range(kw):
                out.append((ky * dil - pad, kx * dil - pad, ky * kw + kx))
        return out

    def add_pack(self, weight, taps_widx, transpose, K_pad, ldw, fmt=0, cin=None):
        """register a pack job; returns the Buf of the packed slab [ntaps][K_pad][ldw] (fmt 0, fp32) or of the split-bf16
        fragments [ntaps][K_pad/16][ldw/32][3][64][8] (fmt 1, 6 bytes per weight; pmf_conv_desc_t.w_s3)."""
        # (fmt 2, the stem class: ONE virtual tap, K_pad = taps * 8 rounded up to 16)
        buf = self.persist.alloc((6 if fmt else 4) * (1 if fmt == 2 else len(taps_widx)) * K_pad * ldw)
        # forward packs: index of the conv op about to be emitted (the first reader); input-gradient packs: none (they are
        # read by the backward graph only)
        owner = len(self.fwd) if (not transpose and self.fwd is not None) else None
        # cin = (first input channel, count): pack that channel range only (its own column origin)
        self.pack_jobs.append((weight, buf, list(taps_widx), transpose, K_pad, ldw, self.lane, fmt, owner, cin))
        return buf

    def s3_ok(self, shape_fill):
        """True when this conv launch may run on the bf16 matrix pipe with split operands (conv_fwd.hip PIPE 5)."""
        if not self.s3:
            return False
        probe = L.ConvDesc()
        shape_fill(probe)
        if probe.ntaps == 1:
            # 1x1 layers: only the direct variant (activations straight from global memory, conv_fwd.hip PIPE 11), and only
            # where the map is large enough to fill the chip without a K split
            return (self.s3_direct_min_pix > 0 and probe.N * probe.OH * probe.OW >= self.s3_direct_min_pix
                    and L.lib().pmf_conv_s3_eligible(C.byref(probe)) == 2)
        if probe.ntaps < self.s3_min_taps:
            return False
        return int(L.lib().pmf_conv_s3_eligible(C.byref(probe)))     # (3: the stem class, weights in pack format 2)

    def conv(self, srcs, conv, act=L.ACT_NONE, bn=None, order="act_bn", relu_view=False, name="", pmask=None,
             extra_bias=None):
        """Conv2d (+bias) -> act -> [BatchNorm]  (order 'act_bn', SalsaNext style) or
        Conv2d -> BatchNorm -> [ReLU on the view]  (order 'bn_act', ResNet / attention style).
        Returns a V.  Registers the backward (BN backward, input gradients, weight gradient)."""
        kh, kw = conv.kernel_size
        dil, pad, stride = conv.dilation[0], conv.padding[0], conv.stride[0]
        Cout = conv.out_channels
        t0 = srcs[0].t
        N = t0.N
        inH = max(s.t.H for s in srcs)
        inW = max(s.t.W for s in srcs)
        OH = (inH + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        OW = (inW + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        out = T(self, N, OH, OW, Cout, name)
        # taps that can only ever read zero padding (|offset| beyond the map: the dilation-12/18 ASPP branches on a
        # 4-row map keep 3 of 9 taps) are dropped from forward, input gradient and weight gradient alike; their weight
        # gradient is exactly zero and stays at the zero the backward prologue writes
        all_taps = self.taps(kh, kw, dil, pad)
        taps = [(dy, dx, wi) for (dy, dx, wi) in all_taps
                if dy < inH and dy + (OH - 1) * stride >= 0 and dx < inW and dx + (OW - 1) * stride >= 0] or all_taps[:1]
        Ktot = sum(_ru(s.t.C, 8) for s in srcs)
        ldw = _ru(Cout, 64)
        # very large dilations: per-tap staging (the halo tile would not fit LDS)
        # (rows and columns separately: the dilation-12 / 18 ASPP branches keep one ROW of three taps on the 4-row map -- a
        # 4 x 56-pixel halo tile, not a 32 x 56 one)
        span_y = max(t[0] for t in taps) - min(t[0] for t in taps)
        span_x = max(t[1] for t in taps) - min(t[1] for t in taps)
        gather = 1 if (8 * stride + span_y) * (32 * stride + span_x) * 80 > 110 * 1024 else 0
        # a BatchNorm module in eval mode inside a training plan (frozen statistics, torch semantics): running statistics in
        # the forward pass, no statistics update, backward through the fixed affine map (dgamma / dbeta still flow)
        train_bn = bn is not None and self.training and bn.training
        has_bias = conv.bias is not None
        k_act = act if order == "act_bn" else L.ACT_NONE
        # EPMF SparseVariantConv: (conv + conv.bias + extra bias) * dilated mask; the mask multiplies after the
        # activation (LeakyReLU(0) = 0 and the mask is 0/1, so act(z*m) == act(z)*m) and before the BN statistics
        bsum = None
        if extra_bias is not None:
            if not has_bias:
                raise NotImplementedError("extra_bias needs a conv bias to add to")
            bsum = self.persist.alloc(4 * Cout)

            def fv(op):
                a = op.u.sm
                a.p[0], a.p[1], a.p[2] = conv.bias.data_ptr(), extra_bias.data_ptr(), bsum.ptr
                a.i[0] = Cout
            self.emit(self.fwd, L.OP_VEC_ADD, fv)

        def shape_fill(d):
            d.N, d.OH, d.OW, d.Cout, d.nsrc = N, OH, OW, Cout, len(srcs)
            for i, s in enumerate(srcs):
                d.src[i].C = _ru(s.t.C, 8)
                d.src[i].ldc, d.src[i].H, d.src[i].W = s.t.ldc, s.t.H, s.t.W
                # flags steer the kernel variant (hence the split-K decision the statistics-row probe must match)
                d.src[i].flags = (L.SRC_RELU if s.relu else 0) | (L.SRC_BCAST if s.bcast else 0)
            d.ntaps = len(taps)
            for i, (dy, dx, _) in enumerate(taps):
                d.tdy[i], d.tdx[i] = dy, dx
            d.in_stride, d.gather = stride, gather
            d.out_sy = d.out_sx = 1
            d.splitk_ws, d.splitk_ws_bytes = 1, SPLITK_BYTES      # non-NULL: same split decision as the real launch
        fwd_s3 = self.s3_ok(shape_fill)
        if fwd_s3 == 3:
            # the 7x7 RGB stem: 8 padded channels x 49 taps, two taps per 16-deep MFMA step (conv_fwd.hip PIPE 14)
            wbuf = self.add_pack(conv.weight, [t[2] for t in taps], 0, _ru(len(taps) * 8, 16), ldw, 2)
        else:
            wbuf = self.add_pack(conv.weight, [t[2] for t in taps], 0, Ktot, ldw, int(bool(fwd_s3)))
        stat_rows = 0
        if train_bn:
            probe = L.ConvDesc()
            shape_fill(probe)
            probe.w_s3 = 1 if fwd_s3 else None
            stat_rows = L.lib().pmf_conv_fwd_stat_rows(C.byref(probe))
            # the autotuner (Plan.autotune) may pick another tile configuration: size the rows for any of them
            max_rows = max(stat_rows, L.lib().pmf_conv_fwd_stat_rows_max(C.byref(probe)))
        stats = self.act.alloc(16 * Cout * max_rows) if train_bn else None   # float64 [rows][2][Cout] partials
        bn_train_flag = int(train_bn)

        lane = self.lane

        def f(op, srcs=srcs):
            d = op.u.conv
            shape_fill(d)
            for i, s in enumerate(srcs):
                self.src_struct(s, d.src[i])
            d.ldw = ldw
            if fwd_s3:
                d.w_s3 = wbuf.ptr
            else:
                d.w = wbuf.ptr
            d.bias = (bsum.ptr if bsum is not None else conv.bias.data_ptr()) if has_bias else None
            d.act = k_act
            d.out, d.out_ldc, d.out_H, d.out_W = out.buf.ptr, out.ldc, OH, OW
            d.out_sy = d.out_sx = 1
            d.stats = stats.ptr if stats is not None else None
            d.ep_pmask = pmask.buf.ptr if pmask is not None else None
            d.splitk_ws, d.splitk_ws_bytes = self.sk_bufs[lane].ptr, self.sk_bufs[lane].nbytes
        self.emit(self.fwd, L.OP_CONV, f)
        conv_fwd_index = len(self.fwd) - 1
        conv_flops = 2.0 * N * OH * OW * Cout * conv.in_channels * len(taps)   # algorithmic (SURVEY.md 8d rule)
        self.meta_fwd[len(self.fwd) - 1] = dict(family="conv_fwd", flops=conv_flops, name=name, shape="%dx%dx%d %d->%d t%d s%d" % (
            N, OH, OW, conv.in_channels, Cout, len(taps), stride))

        view = V(out)
        info = None
        if bn is not None:
            Cb = Cout
            scale, shift = self.persist.alloc(4 * Cb), self.persist.alloc(4 * Cb)
            smean, sinv = self.persist.alloc(4 * Cb), self.persist.alloc(4 * Cb)
            count = float(N * OH * OW)
            if train_bn:
                def fb(op):
                    a = op.u.sm
                    for i, p in enumerate((stats.ptr, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                           bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale.ptr,
                                           shift.ptr, smean.ptr, sinv.ptr)):
                        a.p[i] = p
                    a.f[0], a.f[1], a.f[2] = count, bn.momentum, bn.eps
                    a.i[0], a.i[1] = Cb, stat_rows
                self.emit(self.fwd, L.OP_BN_FINALIZE, fb)
                self._conv_fin[conv_fwd_index] = len(self.fwd) - 1      # its row count follows the conv's tile config
            else:
                def fb(op):
                    a = op.u.sm
                    for i, p in enumerate((bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                           bn.running_var.data_ptr(), scale.ptr, shift.ptr, smean.ptr, sinv.ptr)):
                        a.p[i] = p
                    a.f[0] = bn.eps
                    a.i[0] = Cb
                self.emit(self.fwd, L.OP_BN_EVAL, fb)
            info = dict(mean=smean, invstd=sinv, module=bn)
            view = V(out, scale, shift, relu=relu_view, bn=info)
            if train_bn:
                self.bn_modules.append(bn)
        if name:
            self.views[name] = view
        if not self.training:
            return view

        # ------------------------------------------------------------------ backward
        def backward():
            # partial column sums of dz for the conv-bias gradient: one buffer PER LAYER (the weight-gradient op that
            # folds them runs on the side stream while the main stream already works on the next layer)
            dbr_ld = _ru(Cout, 4)
            dbr = self.act.alloc(COL_ROWS * dbr_ld * 4) if (has_bias and pmask is None) else None
            if bn is not None:
                if getattr(view, "_side", None):
                    if view.gy is None:
                        view.gy = T(self, N, OH, OW, Cout, name + ".gy", ldc=out.ldc)
                    view.gy_written = self._fold_side(view, view.gy, view.gy_written)
                    view._side = None
                    view._gy_last = None
                if view.gy is None:
                    raise RuntimeError("plan: no gradient reached BN output of %s" % name)
                if out.g is None:
                    out.g = T(self, N, OH, OW, Cout, name + ".dz", ldc=out.ldc)
                out.g_written = True
                coef = self.act.alloc(12 * Cout)                  # [3][Cout] per-channel backward coefficients
                dgam, dbet = self.pgrad(bn.weight), self.pgrad(bn.bias)
                gyt, dz = view.gy, out.g
                self._touch(gyt)
                self._touch(dz)
                self.colrows_max = max(self.colrows_max, _ru(Cout, 4))

                lw = getattr(view, "_gy_last", None)
                # small maps (<= 2048 pixels: the 4x128 stage and below, a fifth of the BatchNorm layers): column sums,
                # fold and apply in ONE launch (bn.hip bn_bwd_small_k) instead of three latency-bound ones; the
                # input-gradient epilogue then carries no partial sums either (its hook stays empty)
                small = (os.environ.get("PMF_BN_SMALL", "1") != "0" and Cout % 4 == 0
                         and bool(L.lib().pmf_bn_bwd_small_ok(out.npix, Cout)))
                if small:
                    one_row = has_bias and pmask is None

                    def rs(op):
                        a = op.u.sm
                        ps = (gyt.buf.ptr, out.buf.ptr, info["mean"].ptr, bn.weight.data_ptr(), info["invstd"].ptr,
                              dz.buf.ptr, dbr.ptr if (dbr is not None and one_row) else None, self.pgrad_buf.at(dgam),
                              self.pgrad_buf.at(dbet))
                        for i, p in enumerate(ps):
                            a.p[i] = p
                        a.i[0], a.i[1], a.i[2], a.i[3], a.i[4], a.i[5] = gyt.ldc, out.ldc, Cout, bn_train_flag, k_act, dz.ldc
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_BN_BWD_SMALL, rs)
                    self.note_bytes(self.bwd, "bn_bwd_small", 12.0 * out.npix * Cout)
                    self.grad_done[id(bn.weight)] = self.grad_done[id(bn.bias)] = len(self.bwd) - 1
                    dbias_rows = 1 if one_row else 0
                elif lw is not None and lw["lane"] == lane:
                    # the last writer of gy was an input-gradient launch on this lane: it carried the reduction
                    # (sum gy, sum gy*(a - mean) per tile row) in its epilogue; only the fold is left
                    probe = L.ConvDesc()
                    lw["shape"](probe)
                    nrows = L.lib().pmf_conv_fwd_stat_rows(C.byref(probe))
                    rows = self.act.alloc(16 * Cout * max(nrows, L.lib().pmf_conv_fwd_stat_rows_max(C.byref(probe))))
                    lw["hook"].update(rows=rows, mean=info["mean"])

                    def r1(op):
                        a = op.u.sm
                        ps = (rows.ptr, bn.weight.data_ptr(), info["invstd"].ptr, coef.ptr, self.pgrad_buf.at(dgam),
                              self.pgrad_buf.at(dbet))
                        for i, p in enumerate(ps):
                            a.p[i] = p
                        a.i[0], a.i[1], a.i[2] = Cout, nrows, bn_train_flag
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_BN_BWD_FOLD, r1)
                    self._conv_fold.setdefault(lw["index"], []).append(len(self.bwd) - 1)     # (a multi-destination
                    #                                                      launch feeds one fold per destination)
                else:
                    def r1(op):
                        a = op.u.sm
                        ps = (gyt.buf.ptr, out.buf.ptr, info["mean"].ptr, bn.weight.data_ptr(), info["invstd"].ptr,
                              self.bnpart_bufs[lane].ptr, coef.ptr, self.pgrad_buf.at(dgam), self.pgrad_buf.at(dbet))
                        for i, p in enumerate(ps):
                            a.p[i] = p
                        a.i[0], a.i[1], a.i[2], a.i[3] = gyt.ldc, out.ldc, Cout, bn_train_flag
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_BN_BWD_REDUCE, r1)
                    self.note_bytes(self.bwd, "bn_bwd_reduce", 8.0 * out.npix * Cout)
                if not small:
                    self.grad_done[id(bn.weight)] = self.grad_done[id(bn.bias)] = len(self.bwd) - 1

                def r2(op):
                    a = op.u.sm
                    ps = (gyt.buf.ptr, out.buf.ptr, coef.ptr, info["mean"].ptr, dz.buf.ptr,
                          dbr.ptr if dbr is not None else None)
                    for i, p in enumerate(ps):
                        a.p[i] = p
                    a.i[0], a.i[1], a.i[2], a.i[3], a.i[4], a.i[5] = gyt.ldc, out.ldc, Cout, k_act, dz.ldc, dbr_ld
                    a.l[0] = out.npix
                if not small:
                    self.emit(self.bwd, L.OP_BN_BWD_APPLY, r2)
                    self.note_bytes(self.bwd, "bn_bwd_apply", 12.0 * out.npix * Cout)
                    dbias_rows = L.lib().pmf_col_rows(out.npix, Cout) if (has_bias and pmask is None) else 0
            else:
                dz = self.tgrad(out)
                dbias_rows = 0
                if k_act != L.ACT_NONE or has_bias:
                    def r3(op):
                        a = op.u.sm
                        a.p[0], a.p[1] = dz.buf.ptr, out.buf.ptr
                        a.p[2] = dbr.ptr if dbr is not None else None
                        a.i[0], a.i[1], a.i[2], a.i[3], a.i[4] = dz.ldc, out.ldc, k_act, _ru(Cout, 4), dbr_ld
                        a.l[0] = out.npix
                    self.emit(self.bwd, L.OP_ACT_BWD, r3)
                    self.note_bytes(self.bwd, "act_bwd", 12.0 * out.npix * Cout)
                    dbias_rows = L.lib().pmf_col_rows(out.npix, _ru(Cout, 4)) if (has_bias and pmask is None) else 0
            dz = out.g
            if pmask is not None:
                # d/dz of act(z) * m: the activation derivative above was taken from a = act(z)*m (slope of the
                # masked-out zeros is irrelevant) -- multiply by the mask, then the bias gradients are plain column
                # sums of the masked dz (both bias vectors of SparseVariantConv receive the same gradient)
                def rm(op, dz=dz):
                    a = op.u.sm
                    a.p[0], a.p[1], a.p[2] = dz.buf.ptr, pmask.buf.ptr, dz.buf.ptr
                    a.i[0], a.i[1], a.i[2], a.i[3] = dz.ldc, _ru(Cout, 4), dz.ldc, 0
                    a.l[0] = dz.npix
                self.emit(self.bwd, L.OP_PMASK_MUL_BWD, rm)
                if has_bias:
                    boff = self.pgrad(conv.bias)

                    crows = self.act.alloc(COL_ROWS * _ru(Cout, 4) * 4)      # partial rows of the deterministic column sum

                    def rb(op, dz=dz, boff=boff, crows=crows):
                        a = op.u.sm
                        a.p[0], a.p[1], a.p[2] = dz.buf.ptr, self.pgrad_buf.at(boff), crows.ptr
                        a.i[0], a.i[1], a.i[2] = dz.ldc, Cout, 1
                        a.l[0] = dz.npix
                    self.emit(self.bwd, L.OP_COLSUM, rb)
                    self.grad_done[id(conv.bias)] = len(self.bwd) - 1
                    if extra_bias is not None:
                        eoff = self.pgrad(extra_bias)

                        def re(op, boff=boff, eoff=eoff):
                            a = op.u.sm
                            a.p[0], a.p[1], a.p[2] = self.pgrad_buf.at(boff), None, self.pgrad_buf.at(eoff)
                            a.i[0] = Cout
                        self.emit(self.bwd, L.OP_VEC_ADD, re)
                        self.grad_done[id(extra_bias)] = len(self.bwd) - 1
            self._dgrad(srcs, conv, dz, taps, stride, gather, name)
            self._wgrad(srcs, conv, dz, taps, stride, gather, name, dbias_rows, dbr, dbr_ld)
        self.on_backward(backward)
        return view

    def _dgrad(self, srcs, conv, dz, taps, stride, gather, name):
        """input gradients: the same implicit-GEMM kernel run over dz with transposed weights.
        stride 1:  dX[y] = sum_t dz[y - dy_t] W_t^T                      (one launch per operand)
        stride 2:  y = 2m + py:  dX[y] = sum_{t: (py-dy_t) even} dz[m + (py-dy_t)/2] W_t^T   (one launch per parity)"""
        Cout = conv.out_channels
        Kd = _ru(Cout, 8)                       # K of the input-gradient GEMM (padded channels of dz are zero)
        Cin_tot = sum(_ru(s.t.C, 8) for s in srcs)
        if not any(s.t.needs_grad for s in srcs):
            return
        ldwT = _ru(Cin_tot, 64) + 64
        lane = self.lane
        if stride == 1:
            classes = [(0, 0, [(-dy, -dx, wi) for (dy, dx, wi) in taps])]
        else:
            classes = []
            for py in range(2):
                for px in range(2):
                    sub = [((py - dy) // 2, (px - dx) // 2, wi) for (dy, dx, wi) in taps
                           if (py - dy) % 2 == 0 and (px - dx) % 2 == 0]
                    if sub:
                        classes.append((py, px, sub))
        # split-bf16 launches address the transposed weights by 32-column fragments: every operand that receives a
        # gradient must start on one
        offs, co_ = [], 0
        for s in srcs:
            offs.append(co_)
            co_ += _ru(s.t.C, 8)
        aligned = all(o % 32 == 0 for o, s in zip(offs, srcs) if s.t.needs_grad)
        ref = next(s for s in srcs if s.t.needs_grad)

        def class_probe(sub, py, px):
            def fill(d):
                H, W = ref.t.H, ref.t.W
                d.N = dz.N
                d.OH, d.OW = (H, W) if stride == 1 else ((H - py + 1) // 2, (W - px + 1) // 2)
                d.Cout, d.nsrc = ref.t.C, 1
                sv = d.src[0]
                sv.C, sv.ldc, sv.H, sv.W = Kd, dz.ldc, dz.H, dz.W
                d.ntaps = len(sub)
                for i, (dy, dx, _) in enumerate(sub):
                    d.tdy[i], d.tdx[i] = dy, dx
                d.in_stride, d.gather = 1, gather
            return fill
        if Kd % 16 and self.s3:
            # 20 output channels (the logits heads): K = 24 keeps the launch off the split kernels.  Rounded up to 32 the last
            # eight "channels" of a pixel are the first eight of the next pixel (finite values; past the end of the tensor the
            # buffer range check returns 0) against weight rows that are zero -- they add exactly 0
            Kd8, Kd = Kd, _ru(Cout, 16)
            if not all(self.s3_ok(class_probe(sub, py, px)) for (py, px, sub) in classes):
                Kd = Kd8
        dg_s3 = [Kd % 16 == 0 and self.s3_ok(class_probe(sub, py, px)) for (py, px, sub) in classes]
        # operands that do not start on a 32-column fragment (16 + 64 channels): every operand gets its own transposed
        # pack of its channel range, starting at column 0
        own_packs = not aligned and all(dg_s3) and all(_ru(s.t.C, 8) == s.t.C for s in srcs)
        if not aligned and not own_packs:
            dg_s3 = [False] * len(classes)
        packs = None if own_packs else [self.add_pack(conv.weight, [t[2] for t in sub], 1, Kd, ldwT, int(k3))
                                        for (_, _, sub), k3 in zip(classes, dg_s3)]
        # ---- one launch for all operands of a concatenated input (stride 1, split-bf16 path, every operand a whole number
        # of 32-channel fragments): dz is read ONCE and the output-channel ranges go to the operands' gradient tensors, each
        # with its own epilogue (accumulate / Dropout2d multiplier / ReLU mask / BatchNorm-backward partial sums)
        merged = (stride == 1 and len(srcs) >= 2 and not own_packs and aligned and all(dg_s3) and len(srcs) <= L.MAX_SRC
                  and all(s.t.needs_grad and not s.bcast and s.t.C % 32 == 0 for s in srcs)
                  and dz.N * dz.H * dz.W >= int(os.environ.get("PMF_DGRAD_MERGE_MINPIX", "1024")) and os.environ.get("PMF_DGRAD_MERGE", "1") != "0")
        if merged:
            # torch.cat of one tensor (or one root) twice: two destinations would be ONE gradient buffer, written with
            # accumulate 0 and 1 by workgroups of one launch in unspecified order -- such a concat keeps the per-operand
            # launches (stream order)
            holders = [(s.root() if s.root().bn is not None else s.root().t) for s in srcs]
            merged = len({id(h_) for h_ in holders}) == len(holders)
        if merged:
            # the library's own eligibility rule, on the shape-only descriptor (a mismatch would otherwise only surface as
            # PMF_E_ARG when the plan runs)
            probe = L.ConvDesc()
            probe.N, probe.OH, probe.OW = dz.N, ref.t.H, ref.t.W
            probe.Cout, probe.nsrc = sum(s.t.C for s in srcs), 1
            probe.src[0].C, probe.src[0].ldc, probe.src[0].H, probe.src[0].W = Kd, dz.ldc, dz.H, dz.W
            probe.ntaps = len(classes[0][2])
            for i, (dy, dx, _) in enumerate(classes[0][2]):
                probe.tdy[i], probe.tdx[i] = dy, dx
            probe.in_stride, probe.gather, probe.out_sy, probe.out_sx, probe.w_s3 = 1, gather, 1, 1, 1
            probe.ndst = len(srcs)
            for k_, s in enumerate(srcs):
                probe.dst[k_].C = s.t.C
            merged = bool(L.lib().pmf_conv_multi_ok(C.byref(probe)))
        if merged:
            (_, _, sub), wT = classes[0], packs[0]
            parts = []
            for s in srcs:
                r = s.root()
                tgt, acc = self.grad_of(s)
                parts.append(dict(s=s, r=r, tgt=tgt, acc=acc, relu_x=(r.t if r.relu else None), hook={}))
            Ctot = sum(pt["s"].t.C for pt in parts)

            def shape_only(d, sub=sub, parts=parts, Ctot=Ctot):
                d.N, d.OH, d.OW = dz.N, ref.t.H, ref.t.W
                d.Cout, d.nsrc = Ctot, 1
                sv = d.src[0]
                sv.C, sv.ldc, sv.H, sv.W = Kd, dz.ldc, dz.H, dz.W
                d.ntaps = len(sub)
                for i, (dy, dx, _) in enumerate(sub):
                    d.tdy[i], d.tdx[i] = dy, dx
                d.in_stride, d.gather = 1, gather
                d.out_sy = d.out_sx = 1
                d.w_s3 = 1
                d.ndst = len(parts)
                for k, pt in enumerate(parts):
                    d.dst[k].C = pt["s"].t.C

            def f(op, parts=parts, wT=wT, shape_only=shape_only, ldwT=ldwT):
                d = op.u.conv
                shape_only(d)
                d.src[0].x = dz.buf.ptr
                d.ldw = ldwT
                d.w_s3 = wT.ptr
                d.act = L.ACT_NONE
                d.out_H, d.out_W = ref.t.H, ref.t.W
                d.out, d.out_ldc = parts[0]["tgt"].buf.ptr, parts[0]["tgt"].ldc
                for k, pt in enumerate(parts):
                    e, s, r, tgt = d.dst[k], pt["s"], pt["r"], pt["tgt"]
                    e.out, e.out_ldc, e.accumulate = tgt.buf.ptr, tgt.ldc, pt["acc"]
                    if s.cmul is not None:
                        e.ep_cmul, e.ep_cmul_ld = self.masks_ptr + 4 * s.cmul, s.cmul_ld
                    if pt["relu_x"] is not None:
                        e.ep_relu_x, e.ep_relu_ldc = pt["relu_x"].buf.ptr, pt["relu_x"].ldc
                        e.ep_relu_scale = r.scale.ptr if r.scale is not None else None
                        e.ep_relu_shift = r.shift.ptr if r.shift is not None else None
                    if pt["hook"]:
                        e.stats, e.ep_stat_mean = pt["hook"]["rows"].ptr, pt["hook"]["mean"].ptr
                        if pt["relu_x"] is None:
                            e.ep_relu_x, e.ep_relu_ldc, e.ep_flags = r.t.buf.ptr, r.t.ldc, L.EP_STAT_X_ONLY
            self.emit(self.bwd, L.OP_CONV, f)
            for pt in parts:
                r = pt["r"]
                if r.bn is not None and pt["tgt"] is r.gy and self.bn_bwd_fused:
                    r._gy_last = dict(hook=pt["hook"], index=len(self.bwd) - 1, shape=shape_only, lane=lane)
            self.meta_bwd[len(self.bwd) - 1] = dict(
                family="conv_dgrad", flops=2.0 * dz.N * ref.t.H * ref.t.W * Ctot * Cout * len(sub), name=name,
                shape="%dx%dx%d %d->%s t%d s1" % (dz.N, ref.t.H, ref.t.W, Cout, "+".join(str(pt["s"].t.C) for pt in parts),
                                                  len(sub)))
            return
        coloff = 0
        for s in srcs:
            Cs = _ru(s.t.C, 8)
            if s.t.needs_grad:
                if own_packs:
                    ldwT = _ru(Cs, 64) + 64
                    packs = [self.add_pack(conv.weight, [t[2] for t in sub], 1, Kd, ldwT, 1, cin=(coloff, Cs))
                             for (_, _, sub) in classes]
                r = s.root()
                tmp = None
                if s.bcast:
                    # 1x1 map broadcast over the image: full-size gradient into a temp, then per-sample column sums
                    tmp = T(self, dz.N, dz.H, dz.W, s.t.C, name + ".bcast_tmp")
                    if r.t.g is None:
                        r.t.g = T(self, r.t.N, 1, 1, r.t.C, r.t.name + ".g", arena=self.zero_bwd, ldc=r.t.ldc)
                    r.t.g_written = True
                    self._touch(r.t.g)
                    tgt, acc = tmp, 0
                else:
                    tgt, acc = self.grad_of(s)
                    if stride != 1:
                        # parity classes only touch their own pixels: zero first, then accumulate
                        if not acc:
                            self.fill(self.bwd, tgt.buf, tgt.npix * tgt.ldc)
                        acc = 1
                relu_x = r.t if r.relu else None
                for (py, px, sub), wT, k3 in zip(classes, packs, dg_s3):
                    def shape_only(d, s=s, sub=sub, tgt=tgt, py=py, px=px, k3=k3):
                        H, W = tgt.H, tgt.W
                        d.N = dz.N
                        d.OH, d.OW = (H, W) if stride == 1 else ((H - py + 1) // 2, (W - px + 1) // 2)
                        d.Cout, d.nsrc = s.t.C, 1
                        sv = d.src[0]
                        sv.C, sv.ldc, sv.H, sv.W = Kd, dz.ldc, dz.H, dz.W
                        d.ntaps = len(sub)
                        for i, (dy, dx, _) in enumerate(sub):
                            d.tdy[i], d.tdx[i] = dy, dx
                        d.in_stride, d.gather = 1, gather
                        d.out_sy = d.out_sx = stride
                        d.splitk_ws, d.splitk_ws_bytes = 1, SPLITK_BYTES
                        d.w_s3 = 1 if k3 else None
                    # filled in by the BatchNorm backward of the layer that produced this operand when THIS launch is
                    # the last writer of its output gradient: the launch then also writes the BN-backward partial sums
                    hook = {}

                    def f(op, s=s, r=r, sub=sub, wT=wT, tgt=tgt, acc=acc, coloff=(0 if own_packs else coloff), py=py, px=px,
                          relu_x=relu_x, shape_only=shape_only, hook=hook, k3=k3, ldwT=ldwT):
                        d = op.u.conv
                        shape_only(d)
                        H, W = tgt.H, tgt.W
                        d.src[0].x = dz.buf.ptr
                        d.ldw = ldwT
                        if k3:      # fragment column coloff / 32 (3 planes x 1 KiB each)
                            d.w_s3 = wT.ptr + (coloff // 32) * 3 * 1024
                        else:
                            d.w = wT.at(coloff)
                        d.act = L.ACT_NONE
                        d.out, d.out_ldc, d.out_H, d.out_W = tgt.buf.ptr, tgt.ldc, H, W
                        d.out_sy = d.out_sx = stride
                        d.out_oy, d.out_ox = py, px
                        d.accumulate = acc
                        d.splitk_ws, d.splitk_ws_bytes = self.sk_bufs[lane].ptr, self.sk_bufs[lane].nbytes
                        if s.cmul is not None:
                            d.ep_cmul, d.ep_cmul_ld = self.masks_ptr + 4 * s.cmul, s.cmul_ld
                        if relu_x is not None:
                            d.ep_relu_x, d.ep_relu_ldc = relu_x.buf.ptr, relu_x.ldc
                            d.ep_relu_scale = r.scale.ptr if r.scale is not None else None
                            d.ep_relu_shift = r.shift.ptr if r.shift is not None else None
                        if hook:
                            d.stats, d.ep_stat_mean = hook["rows"].ptr, hook["mean"].ptr
                            if relu_x is None:
                                d.ep_relu_x, d.ep_relu_ldc, d.ep_flags = r.t.buf.ptr, r.t.ldc, L.EP_STAT_X_ONLY
                    self.emit(self.bwd, L.OP_CONV, f)
                    if (stride == 1 and tmp is None and r.bn is not None and tgt is r.gy and self.bn_bwd_fused):
                        r._gy_last = dict(hook=hook, index=len(self.bwd) - 1, shape=shape_only, lane=lane)
                    mh = tgt.H if stride == 1 else (tgt.H - py + 1) // 2
                    mw = tgt.W if stride == 1 else (tgt.W - px + 1) // 2
                    self.meta_bwd[len(self.bwd) - 1] = dict(
                        family="conv_dgrad", flops=2.0 * dz.N * mh * mw * s.t.C * Cout * len(sub), name=name,
                        shape="%dx%dx%d %d->%d t%d s%d" % (dz.N, mh, mw, Cout, s.t.C, len(sub), stride))
                if tmp is not None:
                    crows = self.act.alloc(COL_ROWS * tmp.N * _ru(r.t.g.ldc, 4) * 4)

                    def fc(op, tmp=tmp, g=r.t.g, crows=crows):
                        a = op.u.sm
                        a.p[0], a.p[1], a.p[2] = tmp.buf.ptr, g.buf.ptr, crows.ptr
                        a.i[0], a.i[1], a.i[2] = tmp.ldc, g.ldc, tmp.N
                        a.l[0] = tmp.H * tmp.W
                    self.emit(self.bwd, L.OP_COLSUM, fc)
            coloff += Cs

    def _wgrad(self, srcs, conv, dz, taps, stride, gather, name, dbias_rows=0, dbr=None, dbr_ld=0):
        Cout = conv.out_channels
        goff = self.pgrad(conv.weight)
        kh, kw = conv.kernel_size
        span = max(max(t[0] for t in taps) - min(t[0] for t in taps), max(t[1] for t in taps) - min(t[1] for t in taps))
        wg_gather = 1 if (gather or (3 * stride + 1 + span) * (31 * stride + 1 + span) * 128 > 100 * 1024) else 0

        def shape_fill(d):
            d.N, d.OH, d.OW, d.Cout, d.nsrc = dz.N, dz.H, dz.W, Cout, len(srcs)
            for i, s in enumerate(srcs):
                d.src[i].C = _ru(s.t.C, 8)
                d.src[i].ldc, d.src[i].H, d.src[i].W = s.t.ldc, s.t.H, s.t.W
            d.ntaps = len(taps)
            for i, (dy, dx, wi) in enumerate(taps):
                d.tdy[i], d.tdx[i], d.tap_widx[i] = dy, dx, wi
            d.in_stride, d.gather = stride, wg_gather
            d.Cin_real, d.KHW = conv.in_channels, kh * kw
        probe = L.WgradDesc()
        shape_fill(probe)
        probe.flags = L.WGRAD_S3 if self.s3 else 0       # (the kernel choice, hence the grid, depends on it)
        probe.nsplit = 1
        nsplit = L.lib().pmf_conv_wgrad_nsplit(C.byref(probe))
        probe.nsplit = nsplit
        self.wg_scratch = max(self.wg_scratch, L.lib().pmf_conv_wgrad_workspace(C.byref(probe)))

        def f(op):
            d = op.u.wgrad
            shape_fill(d)
            for i, s in enumerate(srcs):
                self.src_struct(s, d.src[i])
            d.dz, d.dz_ldc = dz.buf.ptr, dz.ldc
            d.partial = self.wg_bufs[lane].ptr
            d.nsplit = nsplit
            d.dw_oihw = self.pgrad_buf.at(goff)
            d.accumulate = 0
            d.flags = L.WGRAD_S3 if self.s3 else 0
            if dbias_rows:
                d.dbias_rows, d.dbias_nrows, d.dbias_ld = dbr.ptr, dbias_rows, dbr_ld
                d.dbias_out = self.pgrad_buf.at(boff)
        boff = self.pgrad(conv.bias) if dbias_rows else None
        home = self.lane
        lane = self._wgrad_lane_of(home) if (self.wgrad_lane and self._wgrad_lane_of(home) != home) else home
        if not (self.wgrad_homes >> home) & 1:
            lane = home
        self.n_wgrad += 1
        meta = dict(family="conv_wgrad", flops=2.0 * dz.N * dz.H * dz.W * Cout * conv.in_channels * len(taps), name=name,
                    shape="%dx%dx%d %d->%d t%d" % (dz.N, dz.H, dz.W, conv.in_channels, Cout, len(taps)))
        ws = None
        if self.flat is not None and self.batch_reds:
            # flat training state (the product path): the partial-slab kernel now, the reduction into OIHW later -- the
            # reductions of RED_BATCH consecutive layers are ONE launch (pmf_conv_wgrad_reduce_multi); every layer keeps
            # its own workspace until then (1.65 GB at 64x2048 bs 2: nothing next to 288 GB)
            ws = self.act.alloc(max(L.lib().pmf_conv_wgrad_workspace(C.byref(probe)), 256))

        def emit_ops():          # on the current lane
            if ws is not None:
                def fb(op, f=f, ws=ws):
                    f(op)
                    op.u.wgrad.partial = ws.ptr
                self.emit(self.bwd, L.OP_WGRAD_PART, fb)
                self.meta_bwd[len(self.bwd) - 1] = meta
                pend = self.pending_reds.setdefault(lane, [])
                pend.append((fb, [conv.weight] + ([conv.bias] if dbias_rows else [])))
                if len(pend) >= self.red_batch:
                    self.flush_reds(lane)
            else:
                # per-tensor gradients (tests, stock DistributedDataParallel): partial slabs, then the reduction into
                # OIHW, back to back on the op's lane through that lane's workspace
                self.emit(self.bwd, L.OP_WGRAD_PART, f)
                self.meta_bwd[len(self.bwd) - 1] = meta
                self.emit(self.bwd, L.OP_WGRAD_RED, f)
                self.grad_done[id(conv.weight)] = len(self.bwd) - 1
                if dbias_rows:
                    self.grad_done[id(conv.bias)] = len(self.bwd) - 1
        if lane != home and self.wgrad_policy == "phase":
            if self._wg_phase_done.get(home) or dz.N * dz.H * dz.W <= self.wgrad_thin_pix:
                if not self._wg_phase_done.get(home):
                    self._wg_phase_done[home] = True
                    self.flush_wgrads(home)
                lane = home
        if lane == home:
            emit_ops()
        else:
            # weight-gradient lane: the ops are DEFERRED and emitted in batches behind ONE event of the home lane (every
            # dz stays alive in the arena until the end of the pass, so running a weight gradient late is always legal)
            q = self._wg_deferred.setdefault(home, [])
            q.append(emit_ops)
            if len(q) >= self.wgrad_batch and self.wgrad_policy != "phase":
                self.flush_wgrads(home)

    def flush_wgrads(self, home, final=False):
        if final or self.wgrad_delay <= 0:
            self._release_armed(home)
        q = self._wg_deferred.pop(home, [])
        if not q:
            return
        ready = self.record_event(self.bwd, lane=home)      # everything the batch reads is complete after this op
        if self.wgrad_delay > 0 and not final:
            self._release_armed(home)
            self._wg_armed[home] = [q, ready, self.wgrad_delay]
            return
        self._emit_batch(home, q, ready)

    def _release_armed(self, home):
        arm = self._wg_armed.pop(home, None)
        if arm is not None:
            self._emit_batch(home, arm[0], arm[1])

    def _emit_batch(self, home, q, ready):
        prev = self.lane
        self.lane = self._wgrad_lane_of(home)
        self.wait_event(self.bwd, ready)
        for fn in q:
            fn()
        self.lane = prev

    def flush_reds(self, lane):
        """emit ONE stage-2 launch for the weight gradients queued on ``lane`` by _wgrad (flat training state only)"""
        pend = self.pending_reds.pop(lane, [])
        if not pend:
            return
        prev, self.lane = self.lane, lane

        def f(op, pend=pend):
            lib = L.lib()
            n = len(pend)
            descs = (L.WgradDesc * n)()
            meta = (C.c_int32 * (8 * n))()
            tmp = L.Op()
            blocks = 0
            for j, (fill, _) in enumerate(pend):
                C.memset(C.addressof(tmp), 0, C.sizeof(tmp))      # fills only set what they use (as on a fresh op)
                fill(tmp)
                C.memmove(C.addressof(descs[j]), C.addressof(tmp.u.wgrad), C.sizeof(L.WgradDesc))
                row = (C.c_int32 * 8)()
                nb = lib.pmf_conv_wgrad_reduce_plan(C.byref(descs[j]), row)
                if nb <= 0:
                    raise RuntimeError("pmf_conv_wgrad_reduce_plan failed: %d" % nb)
                row[0] = blocks
                meta[8 * j:8 * j + 8] = row[:]
                blocks += nb
            jd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
            md = torch.frombuffer(bytearray(bytes(meta)), dtype=torch.uint8).to(self.device)
            self._red_tables.append((jd, md))                  # keep the device tables alive with the plan
            a = op.u.sm
            a.p[0], a.p[1] = jd.data_ptr(), md.data_ptr()
            a.i[0], a.i[1] = n, blocks
        self.emit(self.bwd, L.OP_WGRAD_RED_MULTI, f)
        # data parallelism: the gradients this launch finalises (and everything listed before it) may be all-reduced as
        # soon as it AND the ops emitted so far on the two home lanes have run -- one event each; the engine makes its RCCL
        # side stream wait for them (pmf_plan_event_wait), no cut through the plan
        evs = [self._event_after(self._last_op[(id(self.bwd), lane)])[0]]
        for hl in (0, 1):
            ev = self.record_event(self.bwd, lane=hl)
            if ev is not None:
                evs.append(ev[0])
        self.dp_events.append((len(self.bwd), evs))
        self.lane = prev
        for _, params in pend:
            for p in params:
                self.grad_done[id(p)] = len(self.bwd) - 1

    # ---- element-wise primitives -----------------------------------------------------------------
    def add_act(self, a, b, act, name=""):
        t = a.t
        if self.training and (a.cmul is not None or (b is not None and b.cmul is not None)):
            raise NotImplementedError("add_act: (n,c) multipliers on residual operands have no backward here")
        out = T(self, t.N, t.H, t.W, t.C, name)

        def f(op):
            s = op.u.sm
            self.view_struct(a, s.v[0])
            if b is not None:
                self.view_struct(b, s.v[1])
            s.p[0] = out.buf.ptr
            s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = act, out.ldc, t.H * t.W, int(b is not None), _ru(t.C, 4)
            s.l[0] = out.npix
        self.emit(self.fwd, L.OP_ADD_ACT, f)
        self.note_bytes(self.fwd, "add_act", (12.0 if b is not None else 8.0) * out.npix * t.C)
        if self.training:
            def backward():
                g = self.tgrad(out)
                ga, acca = self.grad_of(a) if a.t.needs_grad else (None, 0)
                gb, accb = self.grad_of(b) if (b is not None and b.t.needs_grad) else (None, 0)

                def fb(op):
                    s = op.u.sm
                    s.p[0], s.p[1] = g.buf.ptr, out.buf.ptr
                    s.p[2] = ga.buf.ptr if ga is not None else None
                    s.p[3] = gb.buf.ptr if gb is not None else None
                    s.i[0], s.i[1], s.i[2] = g.ldc, out.ldc, act
                    s.i[3], s.i[4] = (ga.ldc if ga is not None else 0), acca
                    s.i[5], s.i[6] = (gb.ldc if gb is not None else 0), accb
                    s.i[7] = _ru(t.C, 4)
                    s.l[0] = out.npix
                self.emit(self.bwd, L.OP_ADD_ACT_BWD, fb)
                self.note_bytes(self.bwd, "add_act_bwd", 4.0 * out.npix * t.C * (2 + (ga is not None) * (1 + acca) +
                                                                                 (gb is not None) * (1 + accb)))
            self.on_backward(backward)
        return out

    # ---- per-pixel validity masks (EPMF) ---------------------------------------------------------------
    def pmask_from(self, v):
        """mask = (sum_c |x| != 0) of a view (epmf_net.py:67)."""
        t = v.t
        pm = PM(self, t.N, t.H, t.W)

        def f(op):
            s = op.u.sm
            self.view_struct(v, s.v[0])
            s.p[0] = pm.buf.ptr
            s.i[0], s.i[1] = t.H * t.W, _ru(t.C, 4)
            s.l[0] = t.npix
        self.emit(self.fwd, L.OP_PMASK_FROM, f)
        return pm

    def pmask_pool(self, pm, conv):
        """dilated mask of a SparseVariantConv: max-pool of the zero-padded mask with the conv's geometry (:41-43)."""
        kh, kw = conv.kernel_size
        dil, pad, stride = conv.dilation[0], conv.padding[0], conv.stride[0]
        OH = (pm.H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        OW = (pm.W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        out = PM(self, pm.N, OH, OW)

        def f(op):
            s = op.u.sm
            s.p[0], s.p[1] = pm.buf.ptr, out.buf.ptr
            for i, val in enumerate((pm.N, pm.H, pm.W, kh, kw, dil, pad, stride, OH, OW)):
                s.i[i] = val
        self.emit(self.fwd, L.OP_PMASK_POOL, f)
        return out

    def pmask_mul(self, v, pm, name=""):
        """y = view(v) * mask, materialised; backward gx (+)= gy * mask."""
        t = v.t
        out = T(self, t.N, t.H, t.W, t.C, name)

        def f(op):
            s = op.u.sm
            self.view_struct(v, s.v[0])
            s.p[0], s.p[1] = pm.buf.ptr, out.buf.ptr
            s.i[0], s.i[1], s.i[2] = t.H * t.W, _ru(t.C, 4), out.ldc
            s.l[0] = t.npix
        self.emit(self.fwd, L.OP_PMASK_MUL, f)
        if self.training and t.needs_grad:
            if v.cmul is not None:
                raise NotImplementedError("pmask_mul: (n,c) multiplier on the operand has no backward here")

            def backward():
                g = self.tgrad(out)
                gin, acc = self.grad_of(v)

                def fb(op):
                    s = op.u.sm
                    s.p[0], s.p[1], s.p[2] = g.buf.ptr, pm.buf.ptr, gin.buf.ptr
                    s.i[0], s.i[1], s.i[2], s.i[3] = g.ldc, _ru(t.C, 4), gin.ldc, acc
                    s.l[0] = t.npix
                self.emit(self.bwd, L.OP_PMASK_MUL_BWD, fb)
            self.on_backward(backward)
        else:
            out.needs_grad = False
        return out

    def _pool(self, v, kind_f, kind_b, name, with_idx=False):
        t = v.t
        OH, OW = (t.H - 1) // 2 + 1, (t.W - 1) // 2 + 1
        out = T(self, t.N, OH, OW, t.C, name)
        idx = self.act.alloc(t.N * OH * OW * out.ldc) if (with_idx and self.training) else None

        def f(op):
            s = op.u.sm
            self.view_struct(v, s.v[0])
            s.p[0] = out.buf.ptr
            if with_idx:
                s.p[1] = idx.ptr if idx is not None else None
            s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = t.N, t.H, t.W, _ru(t.C, 4), out.ldc
        self.emit(self.fwd, kind_f, f)
        self.note_bytes(self.fwd, "pool", 4.0 * t.C * (t.npix + out.npix) + (out.npix * t.C if with_idx else 0))
        if self.training and t.needs_grad:
            def backward():
                g = self.tgrad(out)
                gin, acc = self.grad_of(v)

                def fb(op):
                    s = op.u.sm
                    if with_idx:
                        self.view_struct(v, s.v[0])
                        s.p[0], s.p[1], s.p[2] = g.buf.ptr, idx.ptr, gin.buf.ptr
                        s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = g.ldc, t.N, t.H, t.W, _ru(t.C, 4)
                        s.i[5], s.i[6] = gin.ldc, acc
                    else:
                        s.p[0], s.p[2] = g.buf.ptr, gin.buf.ptr
                        s.p[1] = (self.masks_ptr + 4 * v.cmul) if v.cmul is not None else None
                        s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = g.ldc, t.N, t.H, t.W, _ru(t.C, 4)
                        s.i[5], s.i[6], s.i[7] = v.cmul_ld, gin.ldc, acc
                self.emit(self.bwd, kind_b, fb)
                self.note_bytes(self.bwd, "pool_bwd", 4.0 * t.C * (t.npix * (1 + acc) + out.npix))
            self.on_backward(backward)
        return out

    def avgpool(self, v, name=""):
        return self._pool(v, L.OP_AVGPOOL, L.OP_AVGPOOL_BWD, name)

    def maxpool(self, v, name=""):
        return self._pool(v, L.OP_MAXPOOL, L.OP_MAXPOOL_BWD, name, with_idx=True)

    def bilinear(self, v, name=""):
        t = v.t
        out = T(self, t.N, 2 * t.H, 2 * t.W, t.C, name)

        def f(op):
            s = op.u.sm
            self.view_struct(v, s.v[0])
            s.p[0] = out.buf.ptr
            s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = t.N, t.H, t.W, _ru(t.C, 4), out.ldc
        self.emit(self.fwd, L.OP_BILINEAR, f)
        self.note_bytes(self.fwd, "bilinear", 4.0 * t.C * (t.npix + out.npix))
        if self.training:
            def backward():
                g = self.tgrad(out)
                gin, acc = self.grad_of(v)

                def fb(op):
                    s = op.u.sm
                    s.p[0], s.p[1] = g.buf.ptr, gin.buf.ptr
                    s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = g.ldc, t.N, t.H, t.W, _ru(t.C, 4)
                    s.i[5], s.i[6] = gin.ldc, acc
                self.emit(self.bwd, L.OP_BILINEAR_BWD, fb)
                self.note_bytes(self.bwd, "bilinear_bwd", 4.0 * t.C * (t.npix * (1 + acc) + out.npix))
            self.on_backward(backward)
        return out

    def pixel_shuffle(self, v, out_cmul=None, out_cmul_ld=0, name=""):
        t = v.t
        Co = t.C // 4
        out = T(self, t.N, 2 * t.H, 2 * t.W, Co, name)

        def f(op):
            s = op.u.sm
            self.view_struct(v, s.v[0])
            s.p[0] = (self.masks_ptr + 4 * out_cmul) if out_cmul is not None else None
            s.p[1] = out.buf.ptr
            s.i[0], s.i[1], s.i[2], s.i[3], s.i[4], s.i[5] = t.N, t.H, t.W, Co, out_cmul_ld, out.ldc
        self.emit(self.fwd, L.OP_PSHUFFLE, f)
        self.note_bytes(self.fwd, "pixel_shuffle", 8.0 * t.C * t.npix)
        if self.training:
            def backward():
                g = self.tgrad(out)
                gin, acc = self.grad_of(v)

                def fb(op):
                    s = op.u.sm
                    s.p[0] = g.buf.ptr
                    s.p[1] = (self.masks_ptr + 4 * out_cmul) if out_cmul is not None else None
                    s.p[2] = (self.masks_ptr + 4 * v.cmul) if v.cmul is not None else None
                    s.p[3] = gin.buf.ptr
                    for i, x in enumerate((g.ldc, t.N, t.H, t.W, Co, out_cmul_ld, v.cmul_ld, gin.ldc, acc)):
                        s.i[i] = x
                self.emit(self.bwd, L.OP_PSHUFFLE_BWD, fb)
                self.note_bytes(self.bwd, "pixel_shuffle_bwd", 4.0 * t.C * t.npix * (2 + acc))
            self.on_backward(backward)
        return out

    def gate(self, f_v, att_v, pcd, name=""):
        t = pcd
        out = T(self, t.N, t.H, t.W, t.C, name)

        def f(op):
            s = op.u.sm
            self.view_struct(f_v, s.v[0])
            self.view_struct(att_v, s.v[1])
            s.p[0], s.p[1] = pcd.buf.ptr, out.buf.ptr
            s.i[0], s.i[1], s.i[2] = pcd.ldc, out.ldc, _ru(t.C, 4)
            s.l[0] = out.npix
        self.emit(self.fwd, L.OP_GATE, f)
        self.note_bytes(self.fwd, "fusion_gate", 16.0 * t.C * t.npix)
        if self.training:
            def backward():
                g = self.tgrad(out)
                gf, accf = self.grad_of(f_v)
                gatt, _ = self.grad_of(att_v)
                gp, accp = self.grad_of(V(pcd))

                def fb(op):
                    s = op.u.sm
                    self.view_struct(f_v, s.v[0])
                    self.view_struct(att_v, s.v[1])
                    s.p[0], s.p[1], s.p[2], s.p[3] = g.buf.ptr, gf.buf.ptr, gatt.buf.ptr, gp.buf.ptr
                    for i, x in enumerate((g.ldc, gf.ldc, accf, gatt.ldc, gp.ldc, accp, _ru(t.C, 4))):
                        s.i[i] = x
                    s.l[0] = out.npix
                self.emit(self.bwd, L.OP_GATE_BWD, fb)
                self.note_bytes(self.bwd, "fusion_gate_bwd", 4.0 * t.C * t.npix * (7 + accf + accp))
            self.on_backward(backward)
        return out

    def global_mean(self, v, name=""):
        t = v.t
        out = T(self, t.N, 1, 1, t.C, name, arena=self.zero_fwd)

        def f(op):
            s = op.u.sm
            self.view_struct(v, s.v[0])
            s.p[0] = out.buf.ptr
            s.i[0], s.i[1], s.i[2] = t.N, t.H * t.W, _ru(t.C, 4)
        self.emit(self.fwd, L.OP_GMEAN, f)
        if self.training:
            def backward():
                g = self.tgrad(out)
                gin, acc = self.grad_of(v)

                def fb(op):
                    s = op.u.sm
                    s.p[0], s.p[2] = g.buf.ptr, gin.buf.ptr
                    s.p[1] = (self.masks_ptr + 4 * v.cmul) if v.cmul is not None else None
                    for i, x in enumerate((t.N, t.H * t.W, _ru(t.C, 4), v.cmul_ld, gin.ldc, acc)):
                        s.i[i] = x
                self.emit(self.bwd, L.OP_GMEAN_BWD, fb)
            self.on_backward(backward)
        return out

    def softmax_out(self, logits, slot, name=""):
        """logits T -> NCHW probabilities written to an external tensor patched per call (slot index)."""
        t = logits

        def f(op):
            s = op.u.sm
            s.p[0] = t.buf.ptr
            s.p[1] = None   # patched per call
            s.i[0], s.i[1], s.i[2], s.i[3] = t.ldc, t.N, t.H * t.W, t.C
        self.emit(self.fwd, L.OP_SOFTMAX, f)
        self.note_bytes(self.fwd, "softmax", 8.0 * t.C * t.npix)
        self.out_slots[slot] = dict(fwd_index=len(self.fwd) - 1, shape=(t.N, t.C, t.H, t.W))
        if self.training:
            def backward():
                if t.g is None:
                    t.g = T(self, t.N, t.H, t.W, t.C, name + ".dlogits", ldc=t.ldc)
                t.g_written = True
                self._touch(t.g)

                def fb(op):
                    s = op.u.sm
                    s.p[0] = s.p[1] = None   # prob / grad_output patched per call
                    s.p[2] = t.g.buf.ptr
                    s.i[0], s.i[1], s.i[2], s.i[3] = t.N, t.H * t.W, t.C, t.ldc
                self.emit(self.bwd, L.OP_SOFTMAX_BWD, fb)
                self.note_bytes(self.bwd, "softmax_bwd", 12.0 * t.C * t.npix)
                self.out_slots[slot]["bwd_index"] = len(self.bwd) - 1
            self.on_backward(backward)

    def external_grad(self, v):
        """tests: declare that dL/d(view) is supplied from outside (written into the returned T before backward)."""
        g, _ = self.grad_of(v)
        return g

    def input_nchw(self, slot, N, C, H, W, name):
        t = T(self, N, H, W, C, name)
        t.needs_grad = False

        def f(op):
            s = op.u.sm
            s.p[0] = None   # patched per call
            s.p[1] = t.buf.ptr
            s.i[0], s.i[1], s.i[2], s.i[3] = N, C, H * W, t.ldc
        self.emit(self.fwd, L.OP_NCHW2NHWC, f)
        self.note_bytes(self.fwd, "nchw_to_nhwc", 8.0 * N * C * H * W)
        self.in_slots[slot] = len(self.fwd) - 1
        return t

    # ------------------------------------------------------------------ finalisation
    def finalise(self):
        dev = self.device
        # backward ops are emitted by walking the tape in reverse; gradients of the flat parameter buffer and
        # the BN reduction scratch live in zero_bwd (one fill at the start of the backward pass)
        if self.training:
            self._emit_tape()
        self.tape = None
        if self.dry:
            # enough of the final layout for the data-parallel range scheduler (segment_cuts / grad_frontier are pure
            # host logic over op indices: tests/test_ddp_gloo.py runs them on a CPU-only host)
            self.bwd_shift = (1 + (self.flat is not None)) if self.training else 0
            self.n_bwd = (len(self.bwd) + self.bwd_shift) if self.training else 0
            return self
        if self.flat is not None and self.training:
            self.pgrad_buf = ExternalBuf(self.flat.grad)
            self.pgrad_floats = self.flat.grad.numel()
        else:
            self.pgrad_buf = self.zero_bwd.alloc(4 * max(self.pgrad_floats, 64)) if self.training else None
        # scratch shared by the ops of ONE lane (they run in stream order): weight-gradient slabs of the per-tensor path,
        # split-K slabs (small maps only), float64 partial rows of the BatchNorm backward reduction
        nl = self.n_lanes
        self.wg_bufs = [self.act.alloc(max(self.wg_scratch, 256)) for _ in range(nl)] if self.training else None
        self.sk_bufs = [self.act.alloc(SPLITK_BYTES) for _ in range(nl)]
        self.bnpart_bufs = [self.act.alloc(COL_ROWS * 2 * max(self.colrows_max, 4) * 8) for _ in range(nl)]
        for a, zero in ((self.act, False), (self.zero_fwd, True), (self.zero_bwd, True), (self.persist, True)):
            a.materialise(dev, zero)
        self.masks_ptr = self.masks.data_ptr() if self.masks is not None else 0
        lib = L.lib()
        # pack job tables (device), one per lane: every lane re-packs the weights of its own layers at the start of the
        # forward pass (they change every step), concurrently with the other lane's ---------------------------------
        # Only the first PACK_EARLY layers of a lane are packed in front of it; everything else (later layers, all
        # input-gradient packs) goes to ONE table that lane 2 works through while lanes 0 / 1 already run their first
        # layers: the first op of a lane that reads a late-packed weight waits for that launch's event.
        self.pack_tables = []
        PACK_EARLY = int(os.environ.get("PMF_PACK_EARLY", "8"))
        late, late_event = [], None
        groups = []
        for lane in sorted({j[6] for j in self.pack_jobs}):
            mine = [j for j in self.pack_jobs if j[6] == lane]
            fwdj = [j for j in mine if j[8] is not None]
            early = mine
            if PACK_EARLY > 0 and self.n_lanes >= 3 and len(fwdj) > PACK_EARLY + 4 and dev.type == "cuda":
                first_late = fwdj[PACK_EARLY]
                waiter = self.fwd[first_late[8]]
                if not ((waiter[2] >> 8) & 0xff) and (waiter[2] & 3) == lane:      # its wait slot is free
                    if late_event is None:
                        late_event = self.n_events
                        self.n_events += 1
                    waiter[2] |= (late_event + 1) << 8
                    early = fwdj[:PACK_EARLY]
                    keep = {id(j) for j in early}
                    late += [j for j in mine if id(j) not in keep]
            groups.append((lane, early))
        if late:
            groups.append((2, late))
        for lane, mine in groups:
            jobs = (L.PackJob * len(mine))()
            blocks = 0
            for j, (w, buf, tap_idx, transpose, K_pad, ldw, _, fmt, _o, cin) in enumerate(mine):
                Cout, Cin, KHW = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
                J = jobs[j]
                J.w, J.dst = w.data_ptr(), buf.ptr
                if cin is not None:
                    J.w, J.w_ld, Cin = w.data_ptr() + 4 * cin[0] * KHW, Cin, cin[1]
                ct = lib.pmf_pack_tile_ci(Cin, KHW)
                J.Cout, J.Cin, J.KHW, J.ntaps, J.transpose = Cout, Cin, KHW, len(tap_idx), transpose
                J.K_pad, J.ldw, J.CT, J.format = K_pad, ldw, ct, fmt
                J.tiles_ci = (Cin + ct - 1) // ct
                J.block_start = blocks
                for i, ti in enumerate(tap_idx):
                    J.tap_idx[i] = ti
                blocks += J.tiles_ci * ((Cout + 31) // 32)
            dev_tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
            bits = lane
            if late and mine is late:
                bits |= (late_event + 1) << 16          # records the event the first late readers wait for
            self.pack_tables.append((lane, dev_tab, len(mine), blocks, bits))
        self.n_pack_jobs = len(self.pack_jobs)
        self.n_pack_blocks = sum(t[3] for t in self.pack_tables)

        def cost_hint(m):
            """duration estimate of an op for the issue-order simulation of pmf_plan_run (pad_ bits 24-30, units of 4 us):
            matrix work at 100 TFLOP/s fp32-equivalent, everything else at 3 TB/s, 6 us per launch at least"""
            us = 6.0
            if m:
                us = max(us, m.get("flops", 0.0) / 100e12 * 1e6, m.get("bytes", 0.0) / 3e12 * 1e6)
            return max(1, min(127, int(round(us / 4.0))))

        def build(lst, prologue, meta):
            n = len(lst) + len(prologue)
            arr = (L.Op * max(n, 1))()
            k = 0
            for ent in prologue + lst:
                kind, fill = ent[0], ent[1]
                arr[k].kind = kind
                arr[k].pad_ = ent[2] if len(ent) > 2 else 0     # lane / wait / record bits (pmf_amd.h)
                arr[k].pad_ |= cost_hint(meta.get(k - len(prologue)) if k >= len(prologue) else None) << 24
                fill(arr[k])
                k += 1
            return arr, n

        def pack_op(tab, n, blocks):
            def f(op):
                a = op.u.sm
                a.p[0] = tab.data_ptr()
                a.i[0], a.i[1] = n, blocks
            return f

        def zero_arena(arena):
            def z(op):
                a = op.u.sm
                a.p[0], a.f[0], a.l[0] = arena.base, 0.0, arena.size // 4
            return z

        # lane 0 first: a side lane forks from the main stream at its first op and must not wait for lane 0's packing
        pro_f = [(L.OP_PACK, pack_op(tab, n, blocks), bits)
                 for lane, tab, n, blocks, bits in sorted(self.pack_tables, key=lambda t: -t[0])]
        if self.zero_fwd.size:
            pro_f.insert(0, (L.OP_FILL, zero_arena(self.zero_fwd)))
        self.fwd_shift = len(pro_f)
        self.fwd_ops, self.n_fwd = build(self.fwd, pro_f, self.meta_fwd)
        if self.training:
            pro_b = [(L.OP_FILL, zero_arena(self.zero_bwd))]
            if self.flat is not None:
                def zero_flat(op, g=self.flat.grad):
                    a = op.u.sm
                    a.p[0], a.f[0], a.l[0] = g.data_ptr(), 0.0, g.numel()
                pro_b.append((L.OP_FILL, zero_flat))
            self.bwd_shift = len(pro_b)
            self.bwd_ops, self.n_bwd = build(self.bwd, pro_b, self.meta_bwd)
        else:
            self.bwd_ops, self.n_bwd, self.bwd_shift = None, 0, 0
        self.fwd_kinds = [e[0] for e in pro_f + self.fwd]
        self.bwd_kinds = [e[0] for e in (pro_b + self.bwd)] if self.training else []
        self.fwd = self.bwd = None
        self.param_ptrs = [p.data_ptr() for p in self.params]
        if self.device.type == "cuda" and os.environ.get("PMF_AUTOTUNE", "1") != "0":
            self.autotune()
        return self

    # ------------------------------------------------------------------ tile-configuration autotuner
    def force_conv_cfg(self, cfg):
        """tests: run EVERY conv launch of this plan with one tile configuration (BN | MT << 8 | K splits << 16;
        0 = heuristics) -- each configuration the autotuner may choose is pinned against float64 this way."""
        lib = L.lib()
        for ops, n, kinds, shift, fins in ((self.fwd_ops, self.n_fwd, self.fwd_kinds, self.fwd_shift, self._conv_fin),
                                           (self.bwd_ops, self.n_bwd, self.bwd_kinds, self.bwd_shift, self._conv_fold)):
            for k in range(n):
                if kinds[k] != L.OP_CONV:
                    continue
                d = ops[k].u.conv
                d.cfg = cfg
                fin = fins.get(k - shift)
                for fi in (fin if isinstance(fin, list) else ([] if fin is None else [fin])):
                    ops[fi + shift].u.sm.i[1] = lib.pmf_conv_fwd_stat_rows(C.byref(d))
        self._graphs.clear()

    def autotune(self):
        """Every conv launch (forward and input gradient) is timed once per distinct shape with a handful of tile
        configurations (output-channel tile 32/64, 128- or 256-pixel tile, K splits) and keeps the fastest; the
        built-in heuristics are one of the candidates.  Choices are cached process-wide by shape, so two plans of one
        process agree bit for bit.  Measured gains over the heuristics: 0-16 % per layer (tools/sweep_conv.sh)."""
        lib = L.lib()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        failed = C.c_int32(-1)
        import ast
        import os
        cache_file = os.environ.get("PMF_TUNE_CACHE")     # optional: persist / reuse the choices across processes
        if cache_file and os.path.exists(cache_file) and not _TUNED:
            with open(cache_file) as f:
                _TUNED.update(ast.literal_eval(f.read()))
        n_known = len(_TUNED)

        def time_op(ops, k, reps=5):
            for _ in range(2):
                lib.pmf_plan_run_range(C.addressof(ops), k, k + 1, stream, C.byref(failed))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rc = lib.pmf_plan_run_range(C.addressof(ops), k, k + 1, stream, C.byref(failed))
                if rc != 0:
                    return float("inf")
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        def key_of(d):
            return (d.N, d.OH, d.OW, d.Cout, d.nsrc,
                    tuple((d.src[i].C, d.src[i].H, d.src[i].W, d.src[i].flags, bool(d.src[i].scale), bool(d.src[i].cmul))
                          for i in range(d.nsrc)),
                    d.ntaps, tuple(d.tdy[i] for i in range(d.ntaps)), tuple(d.tdx[i] for i in range(d.ntaps)),
                    d.in_stride, d.gather, d.act, d.out_sy, d.out_sx, d.accumulate, bool(d.bias), bool(d.ep_cmul),
                    bool(d.ep_relu_x), bool(d.stats), bool(d.ep_pmask), bool(d.ep_stat_mean), d.ep_flags, bool(d.w_s3),
                    tuple((d.dst[i].C, d.dst[i].accumulate, bool(d.dst[i].ep_relu_x), bool(d.dst[i].stats))
                          for i in range(d.ndst)))

        for ops, n, kinds, shift, fins in ((self.fwd_ops, self.n_fwd, self.fwd_kinds, self.fwd_shift, self._conv_fin),
                                           (self.bwd_ops, self.n_bwd, self.bwd_kinds, self.bwd_shift, self._conv_fold)):
            for k in range(n):
                if kinds[k] != L.OP_CONV:
                    continue
                d = ops[k].u.conv
                key = key_of(d)
                if key not in _TUNED:
                    d.cfg = 0
                    stages = lib.pmf_conv_fwd_kstages(C.byref(d))
                    cands = [0]
                    for bn in ((32, 64) if d.Cout > 32 else (32,)):
                        cands.append(bn | (2 << 8) | (1 << 16))
                        for ks in (1, 2, 4, 8, 16):
                            if ks == 1 or (ks <= stages // 2 and d.N * d.OH * d.OW <= 65536):
                                cands.append(bn | (1 << 8) | (ks << 16))
                    if d.w_s3 and 1 < d.ntaps <= 9 and os.environ.get("PMF_TUNE_DIRECT", "1") != "0":
                        # the direct multi-tap variant (no input tile in LDS, conv_fwd.hip PIPE 13): a candidate wherever the
                        # launch runs on split-bf16 weights
                        for bn in ((32, 64) if d.Cout > 32 else (32,)):
                            for mt in (1, 2):
                                cands.append(bn | (mt << 8) | (1 << 16) | L.CFG_DIRECT_TAPS)
                    best_t, best = float("inf"), 0
                    for cfg in cands:
                        d.cfg = cfg
                        t = time_op(ops, k)
                        if t < best_t * 0.97 or (cfg == 0 and t <= best_t):   # 3 % hysteresis against timing noise
                            best_t, best = min(t, best_t), cfg
                    _TUNED[key] = best
                d.cfg = _TUNED[key]
                fin = fins.get(k - shift)
                for fi in (fin if isinstance(fin, list) else ([] if fin is None else [fin])):
                    ops[fi + shift].u.sm.i[1] = lib.pmf_conv_fwd_stat_rows(C.byref(d))
        # (a tuner over the weight-gradient kernel variant / pixel-split count was measured at 25.44 vs 25.45 ms per
        # step -- no gain over the built-in rules -- and removed.)
        torch.cuda.synchronize(self.device)
        if cache_file and len(_TUNED) != n_known:
            with open(cache_file, "w") as f:
                f.write(repr(_TUNED))

    # ------------------------------------------------------------------ debug readers (tests / tools only)
    def read(self, t):
        """materialised tensor -> torch NCHW copy"""
        x = t.buf.tensor((t.N, t.H, t.W, t.ldc))[..., :t.C]
        return x.permute(0, 3, 1, 2).contiguous()

    def read_view(self, v):
        x = self.read(v.t)
        if v.scale is not None:
            sc = v.scale.tensor((v.t.C,)).view(1, -1, 1, 1)
            sh = v.shift.tensor((v.t.C,)).view(1, -1, 1, 1)
            x = x * sc + sh
        if v.relu:
            x = x.clamp_min(0)
        if v.cmul is not None:
            cm = self.masks[v.cmul:v.cmul + v.t.N * v.cmul_ld].view(v.t.N, v.cmul_ld)[:, :v.t.C]
            x = x * cm[:, :, None, None]
        return x

    def segment_cuts(self, k):
        """op indices that split the backward plan into at most k segments for the data-parallel engine: after every
        segment the gradient ranges that became final are handed to RCCL while the next segment computes.
        Flat training state: a weight gradient is final when the batched stage-2 reduction of its layer group has run
        (OP_WGRAD_RED_MULTI, a handful per pass), so the cuts sit RIGHT BEHIND those ops -- a cut placed a few ops in front
        of one (round 3: segments of equal flops) leaves its whole payload (84 MB of the 146 MB at 64x2048) to the end of
        the pass, fully exposed.  Of more candidates than k - 1 the ones with the largest payload are kept; a reduction in
        the last 2 % of the list is not a cut (nothing left to overlap with).  PMF_DP_CUTS=flops: equal-work segments."""
        if getattr(self, "_cuts", None) is not None and self._cuts[0] == k:
            return self._cuts[1]
        n = self.n_bwd
        kinds = getattr(self, "bwd_kinds", None)
        if not kinds:       # dry plan: the entry list is still there
            kinds = [None] * self.bwd_shift + [e[0] for e in self.bwd]
        cuts = None
        if self.flat is not None and k > 1 and os.environ.get("PMF_DP_CUTS", "reds") != "flops":
            cand = [i + 1 for i in range(n) if kinds[i] == L.OP_WGRAD_RED_MULTI and i + 1 <= n - max(4, n // 50)]
            if cand:
                self.__dict__.pop("_frontiers", None)
                prev, gain = [a for (a, _) in self.flat.ranges], []
                for c in cand:
                    f = self.grad_frontier(c)
                    gain.append(sum(x - p for x, p in zip(f, prev)))
                    prev = f
                keep = sorted(sorted(range(len(cand)), key=lambda j: -gain[j])[:k - 1])
                cuts = [0] + [cand[j] for j in keep if gain[j] > 0] + [n]
        if cuts is None:
            w = [1.0 + self.meta_bwd.get(i - self.bwd_shift, {}).get("flops", 0.0) / 2e9 for i in range(n)]
            tot, acc, cuts = sum(w), 0.0, [0]
            for i in range(n):
                acc += w[i]
                if len(cuts) < k and acc >= tot * len(cuts) / k:
                    cuts.append(i + 1)
            if cuts[-1] != n:
                cuts.append(n)
        self._cuts = (k, cuts)
        return cuts

    def dp_gates(self):
        """[(op_end, [plan events])] in list order: once the events of an entry have fired, every gradient that
        grad_frontier(op_end) reports is final.  The last 2 % of the list is left out (nothing left to overlap with)."""
        n = self.n_bwd
        return [(oe + self.bwd_shift, evs) for (oe, evs) in self.dp_events if oe + self.bwd_shift <= n - max(4, n // 50)]

    def dp_schedule(self):
        """the data-parallel all-reduce schedule of one backward pass as pure data: [(events, [(a, b), ...])] in issue order
        -- float ranges [a, b) of the flat gradient buffer that may be all-reduced once ``events`` have fired -- with a last
        entry (None, ranges) for what only the end of the plan finalises.  Every float appears exactly once."""
        front = [a for (a, _) in self.flat.ranges]
        out = []
        for op_end, evs in self.dp_gates():
            new = self.grad_frontier(op_end)
            todo = [(a, f) for a, f in zip(front, new) if f > a]
            if todo:
                out.append((list(evs), todo))
            front = [max(a, f) for a, f in zip(front, new)]
        out.append((None, [(a, b) for a, (_, b) in zip(front, self.flat.ranges) if b > a]))
        return out

    def grad_frontier(self, op_end):
        """per FlatState group: float offset up to which the gradient buffer is final once ops [0, op_end) have run
        (members are laid out in backward order, so the finished part of a group is a prefix of its range)."""
        cache = self.__dict__.setdefault("_frontiers", {})
        if op_end in cache:
            return cache[op_end]
        out = []
        for (a, b), mem in zip(self.flat.ranges, self.flat.members):
            f = a
            for p in mem:
                d = self.grad_done.get(id(p))
                if d is None or d + self.bwd_shift >= op_end:
                    break
                f = self.flat.offset[id(p)] + (p.numel() + 63) // 64 * 64
            out.append(b if op_end >= self.n_bwd else f)
        cache[op_end] = out
        return out

    def run_profiled(self, what, reps=3):
        """run one pass op by op with a HIP event pair around every launch (on the stream the plan uses, lanes off);
        returns [(op kind name, family or None, algorithmic flops, milliseconds, label, algorithmic bytes)].
        Measurement only."""
        ops, n = (self.fwd_ops, self.n_fwd) if what == "forward" else (self.bwd_ops, self.n_bwd)
        kinds = self.fwd_kinds if what == "forward" else self.bwd_kinds
        meta = self.meta_fwd if what == "forward" else self.meta_bwd
        shift = self.fwd_shift if what == "forward" else self.bwd_shift
        stream = torch.cuda.current_stream(self.device).cuda_stream
        failed = C.c_int32(-1)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        # an event pair around ONE launch also times the dispatch gap (2-3 us against 20-150 us of kernel): the conv and
        # weight-gradient launches (idempotent, or accumulating with a data-independent duration) run `reps` times back
        # to back inside their pair, so that the per-launch figure approaches the kernel duration rocprofv3 reports
        rep_kinds = (L.OP_CONV, L.OP_WGRAD_PART, L.OP_WGRAD)
        lanes = L.lib().pmf_plan_lanes(0)
        try:
            for k in range(n):
                r = reps if kinds[k] in rep_kinds else 1
                evs[k][0].record()
                for _ in range(r):
                    rc = L.lib().pmf_plan_run_range(C.addressof(ops), k, k + 1, C.c_void_p(stream), C.byref(failed))
                evs[k][1].record()
                if rc != 0:
                    raise RuntimeError("pmf_amd %s plan failed at op #%d: code %d" % (what, k, rc))
            torch.cuda.synchronize(self.device)
        finally:
            L.lib().pmf_plan_lanes(lanes)
        out = []
        for k in range(n):
            m = meta.get(k - shift, {})
            r = reps if kinds[k] in rep_kinds else 1
            out.append((L.OP_NAMES.get(kinds[k], "?"), m.get("family"), m.get("flops", 0.0), evs[k][0].elapsed_time(evs[k][1]) / r,
                        m.get("name", "") + ("  [" + m["shape"] + "]" if "shape" in m else ""), m.get("bytes", 0.0)))
        return out

    # ------------------------------------------------------------------ running
    def run(self, ops, n, what, begin=0, end=None, sig=None):
        """launch ops[begin:end) on torch's current stream.  ``sig``: hashable summary of every pointer patched into
        the op array for this call; None = never capture.  With a signature the range is replayed from a hipGraph
        captured for exactly these pointers (first sighting: eager run; second: capture).  The model front end
        (models/pmf_net.py _bind_io) stages inputs, outputs and upstream gradients in plan-owned buffers, so its
        signature is constant and ONE graph per range serves every call, wherever the caller's tensors live."""
        import os
        stream = torch.cuda.current_stream(self.device).cuda_stream
        failed = C.c_int32(-1)
        end = n if end is None else end
        if os.environ.get("PMF_DEBUG_STEP"):   # one op at a time with a sync: localises a faulting kernel
            kinds = self.fwd_kinds if what == "forward" else self.bwd_kinds
            for k in range(begin, end):
                rc = L.lib().pmf_plan_run_range(C.addressof(ops), k, k + 1, C.c_void_p(stream), C.byref(failed))
                torch.cuda.synchronize()
                print("[pmf step] %s #%d %s rc=%d" % (what, k, L.OP_NAMES.get(kinds[k], "?"), rc), flush=True)
                if rc != 0:
                    raise RuntimeError("pmf_amd %s plan failed at op #%d: code %d" % (what, k, rc))
            return
        if sig is not None and os.environ.get("PMF_GRAPH", "1") != "0":
            key = (what, begin, end, sig)
            g = self._graphs.get(key)
            if g is None and self._graph_seen.get(key, 0) >= 1:
                ex = C.c_void_p()
                rc = L.lib().pmf_plan_capture(C.addressof(ops), begin, end, C.byref(ex), C.byref(failed))
                if rc == 0:
                    if len(self._graphs) >= 12:     # bounded: drop the oldest executable graph
                        old = next(iter(self._graphs))
                        L.lib().pmf_graph_destroy(self._graphs.pop(old))
                    g = self._graphs[key] = ex
                else:
                    self._graph_seen[key] = -(1 << 30)   # capture unsupported here: stay eager for this key
            if g is not None:
                rc = L.lib().pmf_graph_launch(g, C.c_void_p(stream))
                if rc != 0:
                    raise RuntimeError("pmf_amd %s plan: hipGraphLaunch failed: code %d" % (what, rc))
                return
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if len(self._graph_seen) > 64:
                self._graph_seen.clear()
        rc = L.lib().pmf_plan_run_range(C.addressof(ops), begin, end, C.c_void_p(stream), C.byref(failed))
        if rc != 0:
            kinds = self.fwd_kinds if what == "forward" else self.bwd_kinds
            kname = L.OP_NAMES.get(kinds[failed.value], "?") if 0 <= failed.value < len(kinds) else "?"
            raise RuntimeError("pmf_amd %s plan failed at op #%d (%s): code %d" % (what, failed.value, kname, rc))
