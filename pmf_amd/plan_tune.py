"""Tile-configuration autotuner of a plan (mixin of pmf_amd.plan.Plan): every conv launch is timed once per distinct shape
with the candidate configurations and the fastest is written into the op."""
import ctypes as C
import os

import torch

from . import _lib as L

_TUNED = {}      # process-wide autotuner cache: conv shape key -> tile configuration (Plan.autotune)
_LOADED = set()  # cache files already merged into _TUNED

# The SHIPPED tile configurations: the tuner's choices for every conv shape of the BASELINE configurations (PMF-R34 at 64x2048
# train / eval, S_B 480x640, the KITTI crop 256x1024, PMF-R50 17 classes 32x1024, EPMF-R34, SalsaNext), measured on an MI355X by
# tools/make_tune_cache.py and committed, so that tests, bench and every rank of a job run the SAME plan for those shapes,
# reproducibly across processes.  PMF_AUTOTUNE: "1" (default) shipped choices + live tuning of shapes the file does not know;
# "cache" shipped choices + the built-in heuristics for unknown shapes (no timing: reproducible; what tests/ run);
# "0" heuristics only; "live" ignores the shipped file (what make_tune_cache.py runs under).
SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950.txt")


def tune_mode():
    m = os.environ.get("PMF_AUTOTUNE", "1")
    return {"0": "off", "cache": "cache", "live": "live"}.get(m, "on")


def _merge(path):
    import ast
    if path and path not in _LOADED and os.path.exists(path):
        _LOADED.add(path)
        with open(path) as f:
            for k, v in ast.literal_eval(f.read()).items():
                _TUNED.setdefault(k, v)


def key_of(d):
    """what the tuner keys a conv launch by: everything that selects a kernel variant or sizes its grid"""
    return (d.N, d.OH, d.OW, d.Cout, d.nsrc,
            tuple((d.src[i].C, d.src[i].H, d.src[i].W, d.src[i].flags, bool(d.src[i].scale), bool(d.src[i].cmul))
                  for i in range(d.nsrc)),
            d.ntaps, tuple(d.tdy[i] for i in range(d.ntaps)), tuple(d.tdx[i] for i in range(d.ntaps)),
            d.in_stride, d.gather, d.act, d.out_sy, d.out_sx, d.accumulate, bool(d.bias), bool(d.ep_cmul),
            bool(d.ep_relu_x), bool(d.stats), bool(d.ep_pmask), bool(d.ep_stat_mean), d.ep_flags, bool(d.w_s3),
            tuple((d.dst[i].C, d.dst[i].accumulate, bool(d.dst[i].ep_relu_x), bool(d.dst[i].stats))
                  for i in range(d.ndst)))


def write_cache(path, table):
    with open(path, "w") as f:
        f.write("{\n" + "".join(" %r: %r,\n" % kv for kv in sorted(table.items(), key=repr)) + "}\n")


class PlanTuneMixin(object):
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
        mode = tune_mode()
        cache_file = os.environ.get("PMF_TUNE_CACHE")     # optional: persist / reuse the choices across processes
        _merge(cache_file)                                # (a job's own file wins over the shipped one: merged first)
        if mode != "live":
            _merge(SHIPPED)
        n_known = len(_TUNED)

        def time_op(ops, k, reps=int(os.environ.get("PMF_TUNE_REPS", "5"))):
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

        for ops, n, kinds, shift, fins in ((self.fwd_ops, self.n_fwd, self.fwd_kinds, self.fwd_shift, self._conv_fin),
                                           (self.bwd_ops, self.n_bwd, self.bwd_kinds, self.bwd_shift, self._conv_fold)):
            for k in range(n):
                if kinds[k] != L.OP_CONV:
                    continue
                d = ops[k].u.conv
                key = key_of(d)
                if key not in _TUNED and mode != "cache":
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
                    if d.w_s3 and os.environ.get("PMF_TUNE_WS", "1") != "0" and lib.pmf_conv_ws_ok(C.byref(d)):
                        # the wave-scheduled N-split kernel (csrc/conv_ws.hip): its default tile count per workgroup, and
                        # explicitly one / two / four 32-channel tiles (cfg bits 26-27) where the channel count allows
                        base = 32 | (1 << 8) | (1 << 16) | L.CFG_WS
                        cands.append(base)
                        for code, nco in ((1, 1), (2, 2), (3, 4)):
                            if d.Cout % (32 * nco) == 0 and d.Cout >= 64:
                                cands.append(base | (code << 26))
                    best_t, best = float("inf"), 0
                    for cfg in cands:
                        d.cfg = cfg
                        t = time_op(ops, k)
                        if t < best_t * 0.97 or (cfg == 0 and t <= best_t):   # 3 % hysteresis against timing noise
                            best_t, best = min(t, best_t), cfg
                    _TUNED[key] = best
                d.cfg = _TUNED.get(key, 0)     # ("cache" mode, unknown shape: the built-in heuristics, no timing)
                fin = fins.get(k - shift)
                for fi in (fin if isinstance(fin, list) else ([] if fin is None else [fin])):
                    ops[fi + shift].u.sm.i[1] = lib.pmf_conv_fwd_stat_rows(C.byref(d))
        # (a tuner over the weight-gradient kernel variant / pixel-split count was measured at 25.44 vs 25.45 ms per
        # step -- no gain over the built-in rules -- and removed.)
        torch.cuda.synchronize(self.device)
        if cache_file and (len(_TUNED) != n_known or not os.path.exists(cache_file)):   # (also when the shipped table knew every shape:
            # the other ranks of a job read THIS file)
            write_cache(cache_file, _TUNED)
