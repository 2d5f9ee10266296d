"""Running a plan (mixin of pmf_amd.plan.Plan): eager ranges and hipGraph replay, the per-op profile, the debug readers, and
the data-parallel range scheduler (segment cuts, gradient frontiers, event gates) -- pure host logic over op indices."""
import ctypes as C
import os

import torch

from . import _lib as L


def _ru4(c):
    return (c + 3) // 4 * 4


class PlanRunMixin(object):
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

    def act_decisions(self, model):
        """debug: the piecewise-linear DECISIONS this plan's last forward pass took, keyed the way a checker keys the same
        sites of the reference's module tree (qualified module names = state-dict prefixes):
            ("lrelu", conv)      bool NCHW, stored conv output > 0                (conv -> LeakyReLU -> BN, salsanext.py:27-33)
            ("relu", conv)       bool NCHW, x * scale + shift > 0 of the view     (conv -> BN -> ReLU, pmf_net.py:20-29)
            ("relu_out", block)  bool NCHW, stored residual sum > 0               (ResNet block output)
            ("maxpool", conv)    int64 [N, C, OH, OW] flat input positions        (the stem's MaxPool2d, pmf_net.py:94)
        Lets a checker run ITS backward pass through the same piecewise-linear function (tests/, bench.py --parity-masked);
        the product never calls this."""
        names = {id(m): n for n, m in model.named_modules()}
        dec = {}
        for conv, name, act, relu_view in self.act_sites:
            q = names.get(id(conv))
            if q is None:
                continue
            if act in (L.ACT_LRELU, L.ACT_RELU):
                dec[("lrelu" if act == L.ACT_LRELU else "relu", q)] = (self.read(self.tensors[name]) > 0).cpu()
            elif relu_view:
                v = self.views[name]
                # the kernels evaluate the view as ONE fused multiply-add (common.h pmf_view_load4 under hipcc's default
                # contraction, conv_ws.hip ws_fma, the epilogue's ReLU mask): its sign is the sign of the exact x * scale + shift,
                # which float64 reproduces (24 x 24-bit product exact, one rounding that cannot cross zero); a float32
                # multiply-then-add rounds twice and disagrees on ~1 element in 10^7 -- one such element is a whole term
                x = self.read(v.t).double()
                x = x * v.scale.tensor((v.t.C,)).double().view(1, -1, 1, 1) + v.shift.tensor((v.t.C,)).double().view(1, -1, 1, 1)
                dec[("relu", q)] = (x > 0).cpu()
        enc = next((n for n, m in model.named_modules() if n.endswith("camera_stream_encoder")), None)
        for name, t in self.tensors.items():
            if enc is not None and name.startswith("enc.layer") and name.endswith(".out"):
                dec[("relu_out", enc + "." + name[4:-4])] = (self.read(t) > 0).cpu()
            idx = getattr(t, "pool_idx", None)
            if idx is not None and enc is not None:
                src = t.pool_of.t
                cq = _ru4(src.C)
                pos = idx.tensor((t.N, t.H, t.W, cq), torch.uint8)[..., :t.C].permute(0, 3, 1, 2).long()
                oy = torch.arange(t.H, device=pos.device).view(1, 1, -1, 1)
                ox = torch.arange(t.W, device=pos.device).view(1, 1, 1, -1)
                y, x = 2 * oy + pos // 3 - 1, 2 * ox + pos % 3 - 1
                dec[("maxpool", enc + ".conv1")] = (y * src.W + x).cpu()
        return dec

    def segment_cuts(self, k):
        """op indices that split the backward plan into at most k segments for the data-parallel engine: after every
        segment the gradient ranges that became final are handed to RCCL while the next segment computes.
        Flat training state: a weight gradient is final when the batched stage-2 reduction of its layer group has run
        (OP_WGRAD_RED_MULTI, a handful per pass), so the cuts sit RIGHT BEHIND those ops -- a cut placed a few ops in front
        of one (round 3: segments of equal flops) leaves its whole payload (84 MB of the 146 MB at 64x2048) to the end of
        the pass, fully exposed.  Of more candidates than k - 1 the ones with the largest payload are kept; a reduction in
        the last 2 % of the list is not a cut (nothing left to overlap with).  Plans without a flat state: equal-work segments."""
        if getattr(self, "_cuts", None) is not None and self._cuts[0] == k:
            return self._cuts[1]
        n = self.n_bwd
        kinds = getattr(self, "bwd_kinds", None)
        if not kinds:       # dry plan: the entry list is still there
            kinds = [None] * self.bwd_shift + [e[0] for e in self.bwd]
        cuts = None
        if self.flat is not None and k > 1:
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

    def pack_ranges(self, ranges, stream):
        """re-pack, on `stream`, the FORWARD-format weights whose parameters lie in the float ranges [a, b) of the flat
        parameter buffer -- the engine calls this right behind the optimiser update of a gradient range (engine.py
        _behind_events), i.e. while the backward plan is still running: the forward pass is over, nothing reads these
        buffers until the next forward.  (The input-gradient packs ARE still being read by the running backward plan: they
        stay in the forward plan's prologue, on lane 2.)  Returns the number of pack jobs launched."""
        key = tuple(ranges)
        cache = self.__dict__.setdefault("_range_packs", {})
        ent = cache.get(key)
        if ent is None:
            base = self.flat.param.data_ptr()
            mine = []
            for j in self.pack_jobs:
                if j[8] is None:
                    continue
                off = (j[0].data_ptr() - base) // 4
                if any(a <= off < b for a, b in ranges):
                    mine.append(j)
            ent = cache[key] = (self._make_pack_table(mine) + (len(mine),)) if mine else (None, 0, 0)
        tab, blocks, n = ent
        if n:
            L.check(L.lib().pmf_pack_weights_batched(tab.data_ptr(), n, blocks, C.c_void_p(stream.cuda_stream)),
                    "pmf_pack_weights_batched")
        return n

    def n_fwd_pack_jobs(self):
        return sum(1 for j in self.pack_jobs if j[8] is not None)

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
