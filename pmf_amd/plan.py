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
from .plan_graph import (_A, SPLITK_BYTES, DBIAS_LD, COL_ROWS, RED_BATCH, _ru, Buf, ExternalBuf, Arena, T, PM, V)  # noqa: F401
from .plan_conv import PlanConvMixin
from .plan_run import PlanRunMixin
from .plan_tune import PlanTuneMixin, _TUNED  # noqa: F401  (tests look the tuner cache up here)


class Plan(PlanConvMixin, PlanTuneMixin, PlanRunMixin):
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
        # split-K launches combine their partial slabs in-kernel (last-arriving workgroup per output tile) instead of through
        # a second launch (conv_finish_k); PMF_SPLITK_FUSED=0: the two-launch form
        self.sk_fused = os.environ.get("PMF_SPLITK_FUSED", "1") != "0"
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
        self.act_sites = []                 # (conv module, debug name, activation, relu-on-view): the piecewise-linear decisions
        self.meta_fwd, self.meta_bwd = {}, {}   # op index (before prologue shift) -> dict(family, flops)
        self.colrows_max = 4

    def knobs(self):
        """the environment knobs this plan was built under (docs/knobs.md documents the defaults; tests/test_host.py
        asserts that a fresh plan in a clean environment carries exactly those)"""
        return {"PMF_CONV_F32": not self.s3, "PMF_S3_MIN_TAPS": self.s3_min_taps, "PMF_S3_DIRECT_MIN_PIX": self.s3_direct_min_pix,
                "PMF_BN_BWD_FUSED": self.bn_bwd_fused, "lanes": self.n_lanes, "PMF_WGRAD_LANE": self.wgrad_lane,
                "PMF_WGRAD_BATCH": self.wgrad_batch, "PMF_RED_BATCH": self.red_batch,
                "PMF_BN_SMALL": os.environ.get("PMF_BN_SMALL", "1") != "0",
                "PMF_DGRAD_MERGE": os.environ.get("PMF_DGRAD_MERGE", "1") != "0",
                "PMF_DGRAD_MERGE_MINPIX": int(os.environ.get("PMF_DGRAD_MERGE_MINPIX", "1024")),
                "PMF_AUTOTUNE": os.environ.get("PMF_AUTOTUNE", "1") != "0",
                "PMF_TUNE_DIRECT": os.environ.get("PMF_TUNE_DIRECT", "1") != "0",
                "PMF_GRAPH": os.environ.get("PMF_GRAPH", "1") != "0",
                "PMF_DP_MODE": os.environ.get("PMF_DP_MODE", "events"),
                "PMF_DP_SEGMENTS": int(os.environ.get("PMF_DP_SEGMENTS", "4")),
                "PMF_PACK_EARLY": int(os.environ.get("PMF_PACK_EARLY", "8"))}

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
        for home in sorted(self._wg_deferred):
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
        out.pool_idx, out.pool_of = idx, v  # (debug: argmax positions 0..8 as uint8 [N, OH, OW, round4(C)])
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
        self.zero_fwd_lanes = getattr(self, "zero_fwd_lanes", set()) | {self.lane}    # (the fill of this arena runs on that lane)

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

    def broadcast(self, small, H, W, name=""):
        """small: T [N, 1, 1, C] -> materialised T [N, H, W, C] with out[n, p, :] = small[n, :] (ASPP's image-level branch:
        `F.interpolate(1x1 -> size)`, pmf_net.py:124-125).  Backward: per-sample column sums of the gradient map
        (pmf_colsum_rows, deterministic) accumulated into the gradient of ``small``."""
        out = T(self, small.N, H, W, small.C, name)

        def f(op):
            s = op.u.sm
            s.p[0], s.p[1] = small.buf.ptr, out.buf.ptr
            s.i[0], s.i[1], s.i[2], s.i[3] = small.ldc, small.N, _ru(small.C, 4), out.ldc
            s.l[0] = H * W
        self.emit(self.fwd, L.OP_BCAST, f)
        self.note_bytes(self.fwd, "broadcast", 4.0 * small.C * out.npix)
        if self.training and small.needs_grad:
            def backward():
                g = self.tgrad(out)
                if small.g is None:
                    small.g = T(self, small.N, 1, 1, small.C, small.name + ".g", arena=self.zero_bwd, ldc=small.ldc)
                small.g_written = True
                self._touch(small.g)
                crows = self.act.alloc(COL_ROWS * small.N * _ru(small.g.ldc, 4) * 4)

                def fc(op):
                    a = op.u.sm
                    a.p[0], a.p[1], a.p[2] = g.buf.ptr, small.g.buf.ptr, crows.ptr
                    a.i[0], a.i[1], a.i[2] = g.ldc, small.g.ldc, small.N
                    a.l[0] = H * W
                self.emit(self.bwd, L.OP_COLSUM, fc)
            self.on_backward(backward)
        return out

    def softmax_out(self, logits, slot, name="", softmax=True):
        """logits T -> NCHW probabilities written to an external tensor patched per call (slot index); softmax=False: the logits
        themselves (SalsaNext(softmax=False), salsanext.py:167,206-207)."""
        t = logits
        ident = int(not softmax)

        def f(op):
            s = op.u.sm
            s.p[0] = t.buf.ptr
            s.p[1] = None   # patched per call
            s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = t.ldc, t.N, t.H * t.W, t.C, ident
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
                    s.i[0], s.i[1], s.i[2], s.i[3], s.i[4] = t.N, t.H * t.W, t.C, t.ldc, ident
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
        # one ticket per output tile for the in-kernel split-K combine (pmf_conv_desc_t.splitk_tickets): zeroed once with the
        # arena, every launch leaves them at zero; one array per lane (the ops of a lane run in stream order)
        self.sk_tickets = [self.persist.alloc(4 * 16384) for _ in range(nl)]
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
        # Tables: per lane the FORWARD-format packs of its first layers ("early"), one table of the later forward packs on
        # lane 2 (its launch records the event the first late readers wait for), and -- behind all of those in list order --
        # the INPUT-GRADIENT packs (read by the backward plan only).  The forward-format tables lead the op list so that a
        # training step whose weights were already re-packed behind the optimiser (engine: pack_ranges) starts the
        # forward range behind them (fwd_pack_skip).
        late, late_event = [], None
        groups, dgroups = [], []
        side_ok = self.n_lanes >= 3 and dev.type == "cuda"
        for lane in sorted({j[6] for j in self.pack_jobs}):
            mine = [j for j in self.pack_jobs if j[6] == lane]
            fwdj = [j for j in mine if j[8] is not None]
            dgj = [j for j in mine if j[8] is None]
            early = fwdj
            if PACK_EARLY > 0 and side_ok and len(fwdj) > PACK_EARLY + 4:
                first_late = fwdj[PACK_EARLY]
                waiter = self.fwd[first_late[8]]
                if not ((waiter[2] >> 8) & 0xff) and (waiter[2] & 3) == lane:      # its wait slot is free
                    if late_event is None:
                        late_event = self.n_events
                        self.n_events += 1
                    waiter[2] |= (late_event + 1) << 8
                    early = fwdj[:PACK_EARLY]
                    late += fwdj[PACK_EARLY:]
            if early:
                groups.append((lane, early))
            if dgj:
                if side_ok:
                    if dgroups and dgroups[0][0] == 2:
                        dgroups[0][1].extend(dgj)
                    else:
                        dgroups.insert(0, (2, list(dgj)))
                else:
                    dgroups.append((lane, dgj))
        if late:
            groups.append((2, late))

        def make_table(mine):
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
            return torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev), blocks

        self._make_pack_table = make_table
        # (list order: forward-format tables with the higher lanes first -- a side lane forks from the main stream at its
        # first op and must not wait for lane 0's packing --, then the input-gradient tables)
        for is_dgrad, grp in ((False, sorted(groups, key=lambda t: -t[0])), (True, dgroups)):
            for lane, mine in grp:
                dev_tab, blocks = make_table(mine)
                bits = lane
                if late and mine is late:
                    bits |= (late_event + 1) << 16          # records the event the first late readers wait for
                self.pack_tables.append((lane, dev_tab, len(mine), blocks, bits, is_dgrad))
        self.fwd_pack_skip = sum(1 for t in self.pack_tables if not t[5])
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

        pro_f = [(L.OP_PACK, pack_op(tab, n, blocks), bits) for lane, tab, n, blocks, bits, _dg in self.pack_tables]
        if self.zero_fwd.size:
            # (behind the pack launches, so that a range starting at fwd_pack_skip still runs it; on the lane of its only
            # users -- the ASPP global mean accumulates into it -- or in front of everything when more than one lane uses it)
            zl = getattr(self, "zero_fwd_lanes", {0})
            if len(zl) == 1:
                pro_f.append((L.OP_FILL, zero_arena(self.zero_fwd), next(iter(zl))))
            else:
                pro_f.insert(0, (L.OP_FILL, zero_arena(self.zero_fwd)))
                self.fwd_pack_skip = 0
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
