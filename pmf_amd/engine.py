"""One PMF optimisation step (tasks/pmf/trainer.py:289-341 of the reference) as a reusable engine.

    normalise LiDAR channels * mask -> PMFNet (HIP plan) -> focal + Lovasz (both heads) + perception-aware loss
    -> backward (HIP plan) -> AdamW(lidar_stream) + SGD-Nesterov(camera encoder+decoder) -> 2 x WarmupCosineLR.step
    -> confusion-matrix update for both heads.

Used by tasks/pmf/trainer.py (real loop) and bench.py (synthetic, device-resident batches) so the benchmark
times exactly what the trainer runs.  The confusion matrices are rank-local until somebody reads a statistic
(IOUEval.getIoU / getAcc / getRecall: one cached all-reduce per read; the reference reduces 6x per iteration, SURVEY.md
5.8) -- the trainer reads at its print frequency and at the end of the epoch, so epoch-end numbers are identical."""
import numpy as np
import os

import torch
import torch.nn as nn

from .loss import (FocalSoftmaxLoss, Lovasz_softmax, MultiTaskLoss, pmf_total_loss, pmf_total_loss_fused,
                   epmf_total_loss, weighted_loss_fused, EPMF_TERMS)
from .metrics import IOUEval
from .utils import WarmupCosineLR


def kitti_focal_alpha(cls_freq, learning_ignore):
    """trainer.py:108-114,194-199: w = 1/(f+1e-3), 0 for ignored; alpha = log(1+w)/max, alpha[0] = 0."""
    w = 1.0 / (np.asarray(cls_freq, np.float64) + 1e-3)
    for c, ig in enumerate(learning_ignore):
        if ig:
            w[c] = 0
    a = np.log(1 + w)
    a = a / a.max()
    a[0] = 0
    return a.astype(np.float32), [c for c in range(len(w)) if w[c] < 1e-10]


class FlatOptimizerView(object):
    """What a trainer sees as ``trainer.optimizer`` when the engine runs on the flat training state: ``step`` /
    ``zero_grad`` / ``param_groups`` are the fused optimiser's own (the LR schedulers act on them), while ``state_dict``
    / ``load_state_dict`` speak the REFERENCE's per-parameter layout (tasks/pmf/main.py:72-83,104-127): parameter ids in
    the order of the reference's parameter lists (trainer.py:80-98: AdamW over lidar_stream.parameters(); SGD over the
    camera encoder's and the decoder's parameters as two groups), one state entry per parameter.  A checkpoint.pth
    written by the reference loads here and vice versa, and the file does not depend on the plan's emission order."""

    _PER_PARAM = ("exp_avg", "exp_avg_sq", "max_exp_avg_sq", "momentum_buffer")

    def __init__(self, optimizer, flat, ref_groups, extra_groups=()):
        """extra_groups: parameter lists that live OUTSIDE the flat buffer as ordinary tensors, one per further
        param_group of the fused optimiser, in order (EPMF: the MultiTaskLoss sigmas, the reference's second AdamW group,
        tasks/epmf/trainer.py:105-109); they follow the flat parameters in the checkpoint, as in the reference."""
        self.optimizer, self.flat = optimizer, flat
        self.ref_groups = [list(g) for g in ref_groups]
        self.extra_groups = [list(g) for g in extra_groups]
        self.flat_param = optimizer.param_groups[0]["params"][0]

    # ---- what the training loop and the schedulers use
    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def step(self, *a, **k):
        return self.optimizer.step(*a, **k)

    def zero_grad(self, set_to_none=True):
        self.flat_param.grad.zero_()        # the gradient buffer is owned by the FlatState: never dropped

    # ---- checkpoint layout
    def _slice(self, t, p):
        base = self.flat.offset[id(p)] - self._group_start
        return t[base:base + p.numel()].view(p.shape)

    @property
    def _group_start(self):
        return min(self.flat.offset[id(p)] for g in self.ref_groups for p in g)

    def state_dict(self):
        st = self.optimizer.state.get(self.flat_param, {})
        state, groups, idx = {}, [], 0
        hyper = {k: v for k, v in self.optimizer.param_groups[0].items() if k != "params"}
        for g in self.ref_groups:
            ids = []
            for p in g:
                ent = {}
                for k, v in st.items():
                    if k in self._PER_PARAM and torch.is_tensor(v) and v.numel() == self.flat_param.numel():
                        ent[k] = self._slice(v, p).clone()
                    elif k == "step":
                        ent[k] = v.clone() if torch.is_tensor(v) else v
                if ent:
                    state[idx] = ent
                ids.append(idx)
                idx += 1
            groups.append(dict(hyper, params=ids))
        for gi, g in enumerate(self.extra_groups):
            hyper_g = {k: v for k, v in self.optimizer.param_groups[1 + gi].items() if k != "params"}
            ids = []
            for p in g:
                ent = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state.get(p, {}).items()}
                if ent:
                    state[idx] = ent
                ids.append(idx)
                idx += 1
            groups.append(dict(hyper_g, params=ids))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        n = sum(len(g) for g in self.ref_groups)
        extra = [p for g in self.extra_groups for p in g]
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != n + len(extra):
            raise ValueError("optimizer checkpoint holds %d parameters, this model has %d" % (len(ids), n + len(extra)))
        for i, p in zip(ids[n:], extra):            # ordinary tensors: their state entries as saved
            ent = sd["state"].get(i, sd["state"].get(str(i)))
            if ent:
                fused = bool(self.optimizer.param_groups[0].get("fused"))
                self.optimizer.state[p] = {
                    k: (v.to(p.device, torch.float32 if k == "step" else v.dtype) if torch.is_tensor(v) and (k != "step" or fused)
                        else (v.clone() if torch.is_tensor(v) else v)) for k, v in ent.items()}
        for gi in range(len(self.extra_groups)):
            src_g = sd["param_groups"][len(self.ref_groups) + gi]
            for k, v in src_g.items():
                if k != "params" and k in self.optimizer.param_groups[1 + gi] and k not in ("fused", "foreach", "capturable"):
                    self.optimizer.param_groups[1 + gi][k] = v
        ids = ids[:n]
        params = [p for g in self.ref_groups for p in g]
        fp = self.flat_param
        st = self.optimizer.state[fp]
        step = None
        for i, p in zip(ids, params):
            ent = sd["state"].get(i, sd["state"].get(str(i)))
            if not ent:
                continue
            for k, v in ent.items():
                if k == "step":
                    step = v if step is None else step
                elif k in self._PER_PARAM and v is not None:
                    if tuple(v.shape) != tuple(p.shape):
                        raise ValueError("optimizer checkpoint: state %r of parameter %d has shape %s, expected %s"
                                         % (k, i, tuple(v.shape), tuple(p.shape)))
                    if k not in st:
                        st[k] = torch.zeros_like(fp.data)
                    self._slice(st[k], p).copy_(v.to(fp.device, fp.dtype))
        if step is not None:
            st["step"] = torch.as_tensor(float(step), dtype=torch.float32, device=fp.device) \
                if self.optimizer.param_groups[0].get("fused") else torch.as_tensor(float(step))
        src = sd["param_groups"][0]
        for k, v in src.items():                # hyper-parameters (lr, betas, momentum, weight decay ...) as saved
            if k != "params" and k in self.optimizer.param_groups[0] and k not in ("fused", "foreach", "capturable"):
                self.optimizer.param_groups[0][k] = v


class TrainEngine:
    def __init__(self, model, nclasses, lr=1e-3, momentum=0.9, weight_decay=1e-5, lambda_=1.0, gamma=0.5, tau=0.7,
                 alpha=None, ignore_class=(0,), warmup_steps=1, max_steps=1, feature_mean=None, feature_std=None,
                 distributed=False, device_ids=None, flat_state=True, adam_weight_decay=None, extra_adam_params=None):
        self.raw_model = model
        dev = next(model.parameters()).device
        self.device = dev
        self.nclasses, self.lambda_, self.gamma, self.tau = nclasses, lambda_, gamma, tau
        fused = dict(fused=True) if dev.type == "cuda" else {}
        self.model = model
        self.flat = None
        self.distributed = distributed
        if dev.type == "cuda" and flat_state:
            # GPU: parameters + gradients re-homed in two flat buffers (models/pmf_net.py FlatState): the backward plan
            # writes p.grad in place, each optimiser is ONE fused launch over one tensor, and data parallelism is a
            # few large all-reduces over contiguous gradient ranges overlapped with the rest of the backward plan
            # (the reference wraps the model in DistributedDataParallel, trainer.py:62-71; same averaged gradients).
            from .models.pmf_net import flatten_training_state
            groups = [list(model.lidar_stream.parameters()),
                      list(model.camera_stream_encoder.parameters()) + list(model.camera_stream_decoder.parameters())]
            self.flat = flatten_training_state(model, groups, dev)
            lidar_params, camera_groups = [self.flat.group_params[0]], [{"params": [self.flat.group_params[1]]}]
            if distributed:
                import torch.distributed as dist
                self._pending, self._frontier = [], None
                self._attach_dp_hooks(model)
                # what DistributedDataParallel does at construction: every rank starts from rank 0's state
                dist.broadcast(self.flat.param, 0)
                for b in model.buffers():
                    dist.broadcast(b, 0)
        else:
            lidar_params = list(model.lidar_stream.parameters())
            camera_groups = [{"params": list(model.camera_stream_encoder.parameters())},      # two groups, as
                             {"params": list(model.camera_stream_decoder.parameters())}]      # trainer.py:88-90
            if distributed:
                self.model = nn.parallel.DistributedDataParallel(
                    model, device_ids=device_ids, gradient_as_bucket_view=True)   # local-stat BN: layers/sync_bn.py
        # trainer.py:80-98: AdamW over the LiDAR stream (torch defaults incl. weight_decay 0.01),
        # SGD-Nesterov over camera encoder + decoder
        adam_groups = [{"params": lidar_params}]
        if extra_adam_params:                               # EPMF: the MultiTaskLoss sigmas (tasks/epmf/trainer.py:105-106)
            adam_groups.append({"params": list(extra_adam_params)})
        adam_kw = {} if adam_weight_decay is None else {"weight_decay": adam_weight_decay}
        self.optimizer = torch.optim.AdamW(adam_groups, lr=lr, **adam_kw, **fused)
        self.aux_optimizer = torch.optim.SGD(camera_groups, lr=lr, nesterov=True, momentum=momentum,
                                             weight_decay=weight_decay, **fused)
        if self.flat is not None:
            # flat state: the updates run as range launches of libpmf_amd.so behind the backward plan's events
            # (_behind_events); the torch optimisers stay as the holders of hyper-parameters and state tensors
            self._setup_range_optim(model, [(self.optimizer, 0, "adamw"), (self.aux_optimizer, 0, "sgd")])
        # what a trainer / main.py holds as trainer.optimizer / trainer.aux_optimizer: reference checkpoint layout
        self.optimizer_view, self.aux_optimizer_view = self.optimizer, self.aux_optimizer
        if self.flat is not None:
            self.optimizer_view = FlatOptimizerView(self.optimizer, self.flat, [list(model.lidar_stream.parameters())])
            self.aux_optimizer_view = FlatOptimizerView(
                self.aux_optimizer, self.flat, [list(model.camera_stream_encoder.parameters()),
                                                list(model.camera_stream_decoder.parameters())])
        if alpha is None:
            alpha = np.ones(nclasses, np.float32)
            alpha[0] = 0
        self.focal = FocalSoftmaxLoss(nclasses, gamma=2, alpha=np.asarray(alpha, np.float32), softmax=False).to(dev)
        self.lovasz = Lovasz_softmax(ignore=0)
        self.metrics = IOUEval(nclasses, dev, ignore=list(ignore_class), is_distributed=distributed)
        self.metrics_img = IOUEval(nclasses, dev, ignore=list(ignore_class), is_distributed=distributed)
        self.scheduler = WarmupCosineLR(self.optimizer, lr, warmup_steps, momentum, max_steps)
        self.aux_scheduler = WarmupCosineLR(self.aux_optimizer, lr, warmup_steps, momentum, max_steps)
        self.mean = None if feature_mean is None else torch.tensor(feature_mean, device=dev).view(1, -1, 1, 1).float()
        self.std = None if feature_std is None else torch.tensor(feature_std, device=dev).view(1, -1, 1, 1).float()
        self.iteration = 0

    # ---- data parallel + optimiser updates over ranges of the flat buffers -------------------------------------
    def _attach_dp_hooks(self, model):
        """PMF_DP_MODE=events (default): the backward plan runs as ONE range and every gradient range is all-reduced behind
        the plan events that finalise it; PMF_DP_MODE=segments: the plan is cut into PMF_DP_SEGMENTS ranges with an
        all-reduce of the finished ranges between them (round 2/3)."""
        if os.environ.get("PMF_DP_MODE", "events") == "segments":
            model._bwd_segment_hook = self._allreduce_ready_ranges
        else:
            model._bwd_gated_hook = self._behind_events

    def _setup_range_optim(self, model, group_opts):
        """group_opts[g] = (torch optimiser, index of its param_group, "adamw" | "sgd") for FlatState group g.
        PMF_OWN_OPTIM=0 keeps the torch fused step() calls at the end of the iteration."""
        self._ro, self._ro_armed, self._ro_done, self._ro_first = None, False, False, {}
        if self.flat is None or self.device.type != "cuda" or os.environ.get("PMF_OWN_OPTIM", "1") == "0":
            return
        for opt, gi, kind in group_opts:
            pg = opt.param_groups[gi]
            if pg.get("maximize") or pg.get("amsgrad") or len(pg["params"]) != 1:
                return                       # not the configurations the reference builds: leave them to torch
        self._ro = list(group_opts)
        if not (self.distributed and os.environ.get("PMF_DP_MODE", "events") == "segments"):
            model._bwd_gated_hook = self._behind_events
        # (segmented all-reduce: the same kernels over the whole buffers once the last collective is in, _finish_range_optim)

    def _arm_range_optim(self):
        """before backward() of a training step: optimiser state tensors exist (torch creates them lazily inside step():
        same names, dtypes and shapes, so state_dict()/load_state_dict() and FlatOptimizerView see no difference) and
        AdamW's step counter counts this step."""
        if self.__dict__.get("_ro") is None:
            return False
        for g, (opt, gi, kind) in enumerate(self._ro):
            fp = opt.param_groups[gi]["params"][0]
            st = opt.state[fp]
            if kind == "adamw":
                if "exp_avg" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=fp.device)
                    st["exp_avg"] = torch.zeros_like(fp, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(fp, memory_format=torch.preserve_format)
                if not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
                    st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=fp.device)
                st["step"] += 1
            else:
                self._ro_first[g] = st.get("momentum_buffer") is None
                if self._ro_first[g]:
                    st["momentum_buffer"] = torch.empty_like(fp, memory_format=torch.preserve_format)
        self._ro_armed, self._ro_done = True, False
        return True

    def _range_update(self, a, b, stream):
        """parameters [a, b) of the flat buffer <- one optimiser step from gradient [a, b) (final, all-reduced), on `stream`"""
        import ctypes as C
        from . import _lib as L
        g = next(i for i, (lo, hi) in enumerate(self.flat.ranges) if lo <= a < hi)
        lo, hi = self.flat.ranges[g]
        self.flat.raw_writes += 1          # (a write torch's version counter does not see: FlatState.stamp)
        if b > hi:
            raise RuntimeError("range optimiser: [%d, %d) crosses the parameter groups" % (a, b))
        opt, gi, kind = self._ro[g]
        pg = opt.param_groups[gi]
        st = opt.state[pg["params"][0]]
        o, n = 4 * (a - lo), b - a
        pp, gp = self.flat.param.data_ptr() + 4 * a, self.flat.grad.data_ptr() + 4 * a
        sp = C.c_void_p(stream.cuda_stream)
        if kind == "adamw":
            rc = L.lib().pmf_adamw_range(pp, gp, st["exp_avg"].data_ptr() + o, st["exp_avg_sq"].data_ptr() + o, n,
                                         float(pg["lr"]), float(pg["betas"][0]), float(pg["betas"][1]), float(pg["eps"]),
                                         float(pg["weight_decay"]), st["step"].data_ptr(), sp)
        else:
            rc = L.lib().pmf_sgd_range(pp, gp, st["momentum_buffer"].data_ptr() + o, n, float(pg["lr"]),
                                       float(pg["momentum"]), float(pg["dampening"]), float(pg["weight_decay"]),
                                       int(bool(pg["nesterov"])), int(self._ro_first[g]), sp)
        if rc != 0:
            raise RuntimeError("libpmf_amd.so: %s range update failed (%d)" % (kind, rc))

    def _behind_events(self, plan):
        """called once per backward pass, right after the whole backward plan has been ENQUEUED (graph replay: ~3 ms of host
        time for ~10 ms of GPU work).  For every batched weight-gradient reduction of the plan (Plan.dp_gates: a handful
        per pass, the first a third of the way in) a side stream waits for the plan events behind it; the gradient ranges
        that are final by then are all-reduced from that stream (data parallel) and, in a training step, the parameters
        of those ranges are updated right behind it -- while the backward plan is still running, and without cutting the
        plan anywhere.  What only the end of the plan finalises follows on the training stream.  Same ranges, same order
        on every rank (the plan is deterministic)."""
        import contextlib
        import ctypes as C
        from . import _lib as L
        dist_on = self.distributed and os.environ.get("PMF_DP_DEBUG_SKIP") != "1"
        armed = self.__dict__.get("_ro") is not None and self._ro_armed and not self._ro_done
        if not dist_on and not armed:
            return
        if dist_on:
            import torch.distributed as dist
        cur = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_dp_side")
        if side is None:
            side = self._dp_side = torch.cuda.Stream(device=self.device)
        lib = L.lib()
        if getattr(self, "time_allreduce", False):
            self._tail_e0 = torch.cuda.Event(enable_timing=True)
            self._tail_e0.record(cur)                    # fires when the backward plan itself is through
        used_side = False
        repack = armed and os.environ.get("PMF_PACK_BEHIND_OPTIM", "0") == "1" and getattr(plan, "fwd_pack_skip", 0) > 0
        packed = 0
        plan.packed_version = None
        for evs, ranges in plan.dp_schedule():
            if evs is not None:
                ok = all(lib.pmf_plan_event_wait(e, C.c_void_p(side.cuda_stream)) == 0 for e in evs)
                if not ok:                  # events never recorded (PMF_LANES=0): order behind the whole plan instead
                    side.wait_stream(cur)
                used_side = True
            where = side if evs is not None else cur          # (None: finalised by the last ops of the plan)
            with (torch.cuda.stream(side) if evs is not None else contextlib.nullcontext()):
                timed = dist_on and getattr(self, "time_allreduce", False)
                if timed:                   # per-range duration of the collective on the stream it is issued from (bench.py)
                    r0 = torch.cuda.Event(enable_timing=True)
                    r0.record(where)
                hs = [dist.all_reduce(self.flat.grad[a:b], async_op=True) for a, b in ranges] if dist_on else []
                if timed:
                    for h in hs:
                        h.wait()
                    r1 = torch.cuda.Event(enable_timing=True)
                    r1.record(where)
                    self.__dict__.setdefault("range_events", []).append((sum(b - a for a, b in ranges), r0, r1))
                if armed:
                    for h in hs:
                        h.wait()            # orders `where` behind the collective (no host block)
                    for a, b in ranges:
                        self._range_update(a, b, where)
                    if repack:
                        # the updated weights of these ranges in the forward plan's GEMM layout, right here: the forward
                        # pass that read the old ones is over, the next one then starts without its pack launches
                        # (opt-in, PMF_PACK_BEHIND_OPTIM=1: measured neutral on the step -- 15.05 ms without, 15.10 ms with, one box;
                        # the "1.1 ms" of the PMF_SKIP_OPS what-if was stale weights, not saved launches: docs/rounds/r05.md)
                        packed += plan.pack_ranges(ranges, where)
                else:
                    self._pending += hs
        if used_side and armed:
            cur.wait_stream(side)
        if armed:
            self._ro_done = True
            if repack and packed == plan.n_fwd_pack_jobs():
                plan.packed_version = self.flat.stamp()       # valid until something else writes the parameters

    def _finish_range_optim(self):
        """after backward(): every parameter range has been updated (or is queued behind its events).  If the hook did
        not run (taken off the model, or a backward pass outside the plan's gated path) the whole buffers are updated
        here on the training stream -- same kernels."""
        if not self.__dict__.get("_ro") or not self._ro_armed:
            return False
        if not self._ro_done:
            cur = torch.cuda.current_stream(self.device)
            for lo, hi in self.flat.ranges:
                if hi > lo:
                    self._range_update(lo, hi, cur)
        self._ro_armed = self._ro_done = False
        e0 = self.__dict__.pop("_tail_e0", None)
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.__dict__.setdefault("allreduce_events", []).append((e0, e1))
        return True

    _allreduce_behind_events = _behind_events        # (round-4 name)

    def _allreduce_ready_ranges(self, plan, op_end):
        """called between segments of the backward plan: all-reduce the gradient ranges that are final by now.
        The collective is asynchronous (RCCL stream): it overlaps with the remaining segments."""
        import torch.distributed as dist
        if os.environ.get("PMF_DP_DEBUG_SKIP") == "1":       # diagnosis only: segments without collectives
            return
        front = plan.grad_frontier(op_end)
        if self._frontier is None:
            self._frontier = [a for (a, _) in self.flat.ranges]
        for g, f in enumerate(front):
            a = self._frontier[g]
            if f > a:
                self._pending.append(dist.all_reduce(self.flat.grad[a:f], async_op=True))
                self._frontier[g] = f

    def _finish_allreduce(self):
        """make the training stream wait for the collectives still in flight (``wait`` orders the streams; it does not
        block the host).  With ``time_allreduce`` set (bench.py) a HIP event pair brackets the waits: the elapsed time
        between them is what the compute stream idled for the all-reduce AFTER the backward plan had finished -- the
        EXPOSED part of the collective, the figure that decides the 1 -> 8 GPU scaling."""
        timed = getattr(self, "time_allreduce", False) and self._pending and self._pending[0] is not None \
            and torch.cuda.is_available() and self.__dict__.get("_tail_e0") is None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self._pending:
            h.wait()
        if timed:
            e1.record()
            self.__dict__.setdefault("allreduce_events", []).append((e0, e1))
        self._pending, self._frontier = [], None

    def exposed_allreduce_ms(self, reset=True):
        """mean exposed all-reduce time per step over the steps recorded since the last reset (see _finish_allreduce)"""
        evs = self.__dict__.get("allreduce_events", [])
        if not evs:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        if reset:
            self.allreduce_events = []
        return ms

    def range_allreduce_us(self, reset=True):
        """[(floats in the range, mean microseconds from issue to completion on the issuing stream)] per gated range of a
        backward pass, over the steps recorded since the last reset -- includes the wait for the plan events in front of the
        collective only insofar as the stream was idle before (bench.py reports it next to the exposed tail)"""
        evs = self.__dict__.get("range_events", [])
        if not evs:
            return None
        torch.cuda.synchronize()
        by = {}
        order = []
        for nfl, a, b in evs:
            if nfl not in by:
                by[nfl] = []
                order.append(nfl)
            by[nfl].append(a.elapsed_time(b) * 1e3)
        if reset:
            self.range_events = []
        return [(nfl, sum(by[nfl]) / len(by[nfl])) for nfl in order]

    def sync_buffers(self):
        """rank 0's BatchNorm running statistics / counters to every rank, as two coalesced broadcasts.
        DistributedDataParallel (broadcast_buffers=True, the reference's default: trainer.py:38-39) does this at every
        forward; since rank 0's buffers are never written by another rank and train-mode BatchNorm normalises with batch
        statistics, doing it once before anything READS the running statistics (validation, checkpoint) leaves every
        observable identical -- eval_step calls it automatically after training steps."""
        self._buffers_dirty = False
        if not self.distributed:
            return
        import torch.distributed as dist
        for want_float in (True, False):
            bufs = [b for b in self.raw_model.buffers() if b.is_floating_point() == want_float]
            if not bufs:
                continue
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, 0)
            torch._foreach_copy_(bufs, [c.view(b.shape) for c, b in zip(flat.split([b.numel() for b in bufs]), bufs)])

    def _step_extra_groups(self):
        """parameters outside the flat buffers (EPMF: the six MultiTaskLoss sigmas in the AdamW's second group) keep the
        torch step: the flat parameter is hidden from it for the call (step() skips parameters without a gradient)."""
        if len(self.optimizer.param_groups) < 2:
            return
        fp = self.optimizer.param_groups[0]["params"][0]
        g, fp.grad = fp.grad, None
        try:
            self.optimizer.step()
        finally:
            fp.grad = g

    def prepare(self, input_feature, input_mask):
        """trainer.py:291-297: normalise the 5 LiDAR channels in place, return the two channel-slice views."""
        if self.mean is not None:
            x, m = input_feature, input_mask
            if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and m.is_cuda and m.dtype == torch.float32
                    and m.is_contiguous() and x.dim() == 4 and x.shape[1] >= 5 and m.shape == (x.shape[0],) + x.shape[2:]):
                # the same arithmetic as one launch of libpmf_amd.so (bit-identical; four ATen passes otherwise)
                import ctypes as C
                from . import _lib as L
                hw = x.shape[2] * x.shape[3]
                L.check(L.lib().pmf_normalise_inplace(x.data_ptr(), x.stride(0), m.data_ptr(), self.mean.data_ptr(),
                                                      self.std.data_ptr(), x.shape[0], 5, hw,
                                                      C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)),
                        "pmf_normalise_inplace")
            else:
                input_feature[:, 0:5] = (input_feature[:, 0:5] - self.mean) / self.std * input_mask.unsqueeze(1)
        return input_feature[:, 0:5], input_feature[:, 5:8]

    def forward_loss(self, pcd, rgb, label):
        """returns (total, terms, lidar_pred, camera_pred, metrics_done).  On the GPU the objective, its gradient and
        the confusion-matrix update of both heads are one fused HIP pass (loss/fused.py); the torch-op version
        (loss/perception.py) serves CPU host tests."""
        lidar_pred, camera_pred = self.model(pcd, rgb)
        if lidar_pred.is_cuda:
            for m in (self.metrics, self.metrics_img):
                if m.conf_matrix.device != lidar_pred.device:
                    m.conf_matrix = m.conf_matrix.to(lidar_pred.device)
            total, terms = pmf_total_loss_fused(lidar_pred, camera_pred, label, self.focal.alpha, self.lambda_,
                                                self.gamma, self.tau, self.focal.gamma, self.metrics.conf_matrix,
                                                self.metrics_img.conf_matrix, grad_out=self._plan_grad_buffers())
            self.metrics.external_update()
            self.metrics_img.external_update()
            return total, terms, lidar_pred, camera_pred, True
        total, terms = pmf_total_loss(lidar_pred, camera_pred, label, self.focal, self.lovasz,
                                      self.lambda_, self.gamma, self.tau)
        return total, terms, lidar_pred, camera_pred, False

    def _plan_grad_buffers(self):
        """the upstream-gradient staging buffers (lidar, camera) of the plan that just ran the training forward pass, for the
        fused objective to write into (saves two N C H W copies per step); None when there is no such plan"""
        plan = getattr(self.model, "_last_plan", None)
        sg = getattr(plan, "stage_g", None) if (plan is not None and plan.training and self.model.training
                                                and torch.is_grad_enabled()) else None
        if not sg or "lidar" not in sg or "camera" not in sg:
            return None
        return (sg["lidar"], sg["camera"])

    def train_step(self, input_feature, input_mask, input_label):
        """one full iteration; everything stays on the device (no .item())."""
        self.model.train()
        self._buffers_dirty = True
        pcd, rgb = self.prepare(input_feature, input_mask)
        label = input_label.long()
        total, terms, lidar_pred, camera_pred, metrics_done = self.forward_loss(pcd, rgb, label)
        if self.flat is None:
            self.optimizer.zero_grad(set_to_none=True)
            self.aux_optimizer.zero_grad(set_to_none=True)
        own = self._arm_range_optim()
        if self.flat is not None and self.distributed:
            # mean over ranks = sum of gradients of loss / world: scale the upstream gradient (free: the objective's
            # backward multiplies by it anyway) instead of a pass over the 146 MB gradient buffer after the all-reduce
            import torch.distributed as dist
            total.backward(torch.full_like(total, 1.0 / dist.get_world_size()))
            self._finish_allreduce()
        else:
            from .loss import fused as _fused
            _fused.UNIT_UPSTREAM = True       # (d total / d total = 1: the objective's backward hands its gradient maps on as they are)
            try:
                total.backward()    # flat state: the plan zero-fills and rewrites the gradient buffer itself
            finally:
                _fused.UNIT_UPSTREAM = False
        if own:
            self._finish_range_optim()
            self._step_extra_groups()
        else:
            self.optimizer.step()
            self.aux_optimizer.step()
        self.scheduler.step()
        self.aux_scheduler.step()
        if not metrics_done:
            with torch.no_grad():
                self.metrics.addBatch(lidar_pred.argmax(dim=1), label)
                self.metrics_img.addBatch(camera_pred.argmax(dim=1), label)
        self.iteration += 1
        return total.detach(), terms

    @torch.no_grad()
    def eval_step(self, input_feature, input_mask, input_label):
        self.model.eval()
        if self.distributed and getattr(self, "_buffers_dirty", True) and self.flat is not None:
            self.sync_buffers()
        pcd, rgb = self.prepare(input_feature, input_mask)
        label = input_label.long()
        total, terms, lidar_pred, camera_pred, metrics_done = self.forward_loss(pcd, rgb, label)
        if not metrics_done:
            self.metrics.addBatch(lidar_pred.argmax(dim=1), label)
            self.metrics_img.addBatch(camera_pred.argmax(dim=1), label)
        return total, terms


class EPMFEngine(TrainEngine):
    """One EPMF optimisation step (tasks/epmf/trainer.py:27-33,95-122,340-437 of the reference, PMFNet branch with
    ``use_mtloss``): six separately weighted terms [foc_img, lov_img, per_img, per, foc, lov] through MultiTaskLoss(6)
    (learned sigmas, optimised by the AdamW that also owns the LiDAR stream, weight_decay = settings.weight_decay),
    SGD-Nesterov over the camera stream.  On the GPU the six terms, the gradient of their sigma-weighted sum w.r.t. both
    probability maps and the confusion matrices are one fused HIP pass (loss/fused.py weighted_loss_fused: the weights
    1 / (2 sigma_i^2) are read on the device); the sigma gradients come from autograd through those weights.  Under data
    parallelism the sigma gradient is averaged like every other gradient (the reference wraps mt_loss in its own
    DistributedDataParallel, :44-49)."""

    def __init__(self, model, nclasses, weight_decay=1e-5, **kw):
        dev = next(model.parameters()).device
        self.mt_loss = MultiTaskLoss(6).to(dev)
        super().__init__(model, nclasses, weight_decay=weight_decay, adam_weight_decay=weight_decay,
                         extra_adam_params=self.mt_loss.parameters(), **kw)
        if self.distributed:
            import torch.distributed as dist
            dist.broadcast(self.mt_loss.sigma.data, 0)
            if self.flat is None:
                # per-tensor path (DistributedDataParallel over the model): the sigmas are not inside that wrapper --
                # average their gradient as the reference's own DistributedDataParallel(mt_loss) does (:44-49)
                world = dist.get_world_size()

                def _avg(p):
                    dist.all_reduce(p.grad)
                    p.grad.div_(world)
                self.mt_loss.sigma.register_post_accumulate_grad_hook(_avg)
        if self.flat is not None:
            # checkpoint view in the reference's layout: the lidar parameters one by one, then the sigmas as group 2
            self.optimizer_view = FlatOptimizerView(self.optimizer, self.flat, [list(model.lidar_stream.parameters())],
                                                    extra_groups=[list(self.mt_loss.parameters())])

    def forward_loss(self, pcd, rgb, label):
        if self.flat is not None:
            self.mt_loss.sigma.grad = None              # (the flat buffer is re-zeroed by the backward plan itself)
        lidar_pred, camera_pred = self.model(pcd, rgb)
        if lidar_pred.is_cuda:
            for m in (self.metrics, self.metrics_img):
                if m.conf_matrix.device != lidar_pred.device:
                    m.conf_matrix = m.conf_matrix.to(lidar_pred.device)
            sg2 = self.mt_loss.sigma.pow(2)
            w = 1.0 / (2.0 * sg2)                                             # multi_task_loss.py:17-18
            # fused term order (foc, lov, foc_cam, lov_cam, per, per_img) <- sigma order of EPMF_TERMS
            order = [EPMF_TERMS.index(k) for k in ("foc", "lov", "foc_cam", "lov_cam", "per", "per_img")]
            total, terms = weighted_loss_fused(lidar_pred, camera_pred, label, self.focal.alpha, w[order], self.tau,
                                               self.focal.gamma, self.metrics.conf_matrix, self.metrics_img.conf_matrix,
                                               grad_out=self._plan_grad_buffers())
            total = total + (sg2 + 1.0).log().sum()
            self.metrics.external_update()
            self.metrics_img.external_update()
            return total, terms, lidar_pred, camera_pred, True
        total, terms = epmf_total_loss(lidar_pred, camera_pred, label, self.focal, self.lovasz, self.mt_loss, self.tau)
        return total, terms, lidar_pred, camera_pred, False

    def _finish_allreduce(self):
        super()._finish_allreduce()
        if self.distributed and self.mt_loss.sigma.grad is not None:
            import torch.distributed as dist
            dist.all_reduce(self.mt_loss.sigma.grad)      # (1 / world rides on the upstream gradient already)


class SalsaNextEngine:
    """One optimisation step of the LiDAR-only task (tasks/salsanext/trainer.py:171-274 of the reference):

        label <- label * (label >= 1), mask <- mask * (label >= 1) -> SalsaNext (HIP plan) -> Lovasz(ignore 0) +
        focal(gamma 2, alpha, mask) -> backward (HIP plan) -> AdamW(lr) -> WarmupCosineLR.step -> confusion matrix.

    Parameters and gradients live in one flat buffer (one fused AdamW launch); the reference runs this task under
    nn.DataParallel -- here one process per GPU with the same range all-reduce as TrainEngine when distributed."""

    def __init__(self, model, nclasses, lr=1e-3, momentum=0.9, alpha=None, ignore_class=(0,), warmup_steps=1, max_steps=1,
                 distributed=False, flat_state=True):
        dev = next(model.parameters()).device
        self.model, self.device, self.nclasses, self.distributed = model, dev, nclasses, distributed
        fused = dict(fused=True) if dev.type == "cuda" else {}
        self.flat = None
        params = list(model.parameters())
        if dev.type == "cuda" and flat_state:
            from .models.pmf_net import flatten_training_state
            self.flat = flatten_training_state(model, [params], dev)
            params = [self.flat.group_params[0]]
            if distributed:
                import torch.distributed as dist
                self._pending, self._frontier = [], None
                self._attach_dp_hooks(model)
                dist.broadcast(self.flat.param, 0)
                for b in model.buffers():
                    dist.broadcast(b, 0)
        elif distributed:
            raise RuntimeError("SalsaNextEngine: data parallelism needs the flat training state on a GPU")
        self.optimizer = torch.optim.AdamW(params, lr=lr, **fused)                      # trainer.py:57-61
        # what the task script saves / restores (tasks/salsanext/main.py:64-70,105-110): per-parameter layout
        self.optimizer_view = self.optimizer if self.flat is None else \
            FlatOptimizerView(self.optimizer, self.flat, [list(model.parameters())])
        if alpha is None:
            alpha = np.ones(nclasses, np.float32)
            alpha[0] = 0
        self.focal = FocalSoftmaxLoss(nclasses, gamma=2, alpha=np.asarray(alpha, np.float32), softmax=False).to(dev)
        self.lovasz = Lovasz_softmax(ignore=0)
        self.metrics = IOUEval(nclasses, dev, ignore=list(ignore_class), is_distributed=distributed)
        self.scheduler = WarmupCosineLR(self.optimizer, lr, warmup_steps, momentum, max_steps)
        self.iteration = 0
        if self.flat is not None:
            self._setup_range_optim(model, [(self.optimizer, 0, "adamw")])

    _allreduce_ready_ranges = TrainEngine._allreduce_ready_ranges
    _behind_events = TrainEngine._behind_events
    _attach_dp_hooks = TrainEngine._attach_dp_hooks
    _finish_allreduce = TrainEngine._finish_allreduce
    exposed_allreduce_ms = TrainEngine.exposed_allreduce_ms
    _setup_range_optim = TrainEngine._setup_range_optim
    _arm_range_optim = TrainEngine._arm_range_optim
    _range_update = TrainEngine._range_update
    _finish_range_optim = TrainEngine._finish_range_optim

    def forward_loss(self, feature, label, mask):
        """(feature, label, mask): the order SalsaNextLoader yields and the reference loop unpacks
        (salsanext_loader.py:84, tasks/salsanext/trainer.py:186) -- ``eng.train_step(*batch)`` is right by construction."""
        label = label.long()
        label = label * label.ge(1).long()
        mask = mask.to(feature.dtype) * label.ge(1).to(feature.dtype)
        output = self.model(feature)
        loss_s = self.focal(output, label, mask=mask)
        loss_lovasz = self.lovasz(output, label)
        return loss_lovasz + loss_s, {"focal": loss_s.detach(), "lovasz": loss_lovasz.detach()}, output, label

    def train_step(self, feature, label, mask):
        self.model.train()
        total, terms, output, label = self.forward_loss(feature, label, mask)
        if self.flat is None:
            self.optimizer.zero_grad(set_to_none=True)
        own = self._arm_range_optim()
        if self.flat is not None and self.distributed:
            import torch.distributed as dist
            total.backward(torch.full_like(total, 1.0 / dist.get_world_size()))
            self._finish_allreduce()
        else:
            total.backward()
        if own:
            self._finish_range_optim()
        else:
            self.optimizer.step()
        self.scheduler.step()
        with torch.no_grad():
            self.metrics.addBatch(output.argmax(dim=1), label)
        self.iteration += 1
        return total.detach(), terms

    @torch.no_grad()
    def eval_step(self, feature, label, mask):
        self.model.eval()
        total, terms, output, label = self.forward_loss(feature, label, mask)
        self.metrics.addBatch(output.argmax(dim=1), label)
        return total, terms
