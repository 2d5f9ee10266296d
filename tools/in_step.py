#!/usr/bin/env python3
"""The roofline of the step that was TIMED: one replayed four-lane training step of bench.py seen by
rocprofv3 -- wall time, kernels in flight, and, from the separate PMC passes of tools/collect_profiles.sh, the matrix-pipe
busy share and the HBM bytes of EVERY kernel of the step over that wall time.

usage: in_step.py trace.db mfma.db fetch.db write.db out.json commit bench_plain.json
  trace.db  rocprofv3 --kernel-trace of the default replay (no counters: kernels overlap as in the timed run)
  mfma.db   --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ...   fetch.db / write.db  --pmc FETCH_SIZE / --pmc WRITE_SIZE
Steps are delimited by the once-per-step loss_pixel_k launch; the second-to-last full step is used everywhere.  Counter
passes serialise the kernels, so their per-kernel totals do not depend on the overlap; the wall time comes from the trace
WITHOUT counters (and, beside it, from the unprofiled bench line: tracing itself stretches the step).
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md), WRITE_SIZE uncorrected."""
import json
import sqlite3
import sys


def tabs(c):
    t = lambda p: [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like '%s%%'" % p)][0]
    return t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")


def one_step(db):
    c = sqlite3.connect(db)
    kd, ks = tabs(c)
    rows = c.execute("select d.id, d.start, d.end, s.kernel_name, d.event_id from %s d join %s s on d.kernel_id = s.id "
                     "order by d.start" % (kd, ks)).fetchall()
    marks = [i for i, r in enumerate(rows) if "loss_pixel_k" in r[3]]
    a, b = marks[-3], marks[-2]
    return c, rows[a:b]


def counters(db, names):
    c, step = one_step(db)
    t = lambda p: [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like '%s%%'" % p)][0]
    pe, pi = t("rocpd_pmc_event"), t("rocpd_info_pmc")
    ev = {r[4]: r for r in step}
    out = {n: {} for n in names}
    q = "select e.event_id, p.name, sum(e.value) from %s e join %s p on e.pmc_id = p.id group by e.event_id, p.name" % (pe, pi)
    for eid, name, v in c.execute(q):
        if eid in ev and name in out:
            out[name][eid] = (v, ev[eid][3], ev[eid][2] - ev[eid][1])
    return out, step


def main(trace, mfma, fetch, write, out_path, commit, bench_plain):
    _, step = one_step(trace)
    t0, t1 = step[0][1], max(r[2] for r in step)
    wall_ns = t1 - t0
    evs = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
    hist, depth, prev = [0.0] * 5, 0, evs[0][0]
    for x, k in evs:
        hist[min(depth, 4)] += x - prev
        prev, depth = x, depth + k
    is_conv = lambda n: "conv_fwd_k" in n or "conv_ws_k" in n or "conv_finish_k" in n
    is_mat = lambda n: is_conv(n) or "wgrad" in n
    busy_ns = sum(r[2] - r[1] for r in step)
    res = {"commit": commit, "kernels_in_step": len(step), "step_wall_ms_under_trace": wall_ns / 1e6,
           "sum_of_kernel_durations_ms": busy_ns / 1e6,
           "wall_ms_by_kernels_in_flight": {("%d%s" % (k, "+" if k == 4 else "")): hist[k] / 1e6 for k in range(5)},
           "conv_fwd_k_in_step": {"launches": sum(1 for r in step if "conv_fwd_k" in r[3] or "conv_ws_k" in r[3]),
                                  "avg_launch_us_incl_finish": sum(r[2] - r[1] for r in step if is_conv(r[3])) / 1e3 /
                                  max(1, sum(1 for r in step if "conv_fwd_k" in r[3] or "conv_ws_k" in r[3]))}}
    try:
        with open(bench_plain) as f:
            res["ms_per_step_unprofiled"] = json.loads(f.read().strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        res["ms_per_step_unprofiled"] = None
    wall_ms = res["ms_per_step_unprofiled"] or wall_ns / 1e6
    # ---- matrix pipe: sum over the step's kernels of MFMA-busy cycles / (32 SIMDs of the sampled shader engine x cycles of the step)
    cm, _ = counters(mfma, ("SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"))
    busy, mf = cm["SQ_BUSY_CYCLES"], cm["SQ_VALU_MFMA_BUSY_CYCLES"]
    big = [(v, d) for e, (v, n, d) in busy.items() if is_conv(n) and d > 30000]
    clk_ghz = (sum(v for v, _ in big) / max(1.0, sum(d for _, d in big))) if big else 0.0     # SQ busy cycles per ns of a long kernel
    mfma_total = sum(v for v, _, _ in mf.values())
    mfma_mat = sum(v for v, n, _ in mf.values() if is_mat(n))
    sq_mat = sum(v for e, (v, n, _) in busy.items() if is_mat(n))
    res["matrix_pipe"] = {
        "sq_clock_ghz_from_long_conv_launches": clk_ghz,
        "mfma_busy_over_step": mfma_total / (32.0 * clk_ghz * wall_ms * 1e6) if clk_ghz else None,
        "mfma_busy_inside_matrix_kernels": mfma_mat / (32.0 * sq_mat) if sq_mat else None,
        "note": "SQ_VALU_MFMA_BUSY_CYCLES summed over every kernel of the step / (32 SIMDs x SQ clock x step wall time); "
                "'inside' = the same sum over conv / weight-gradient kernels / their own SQ_BUSY_CYCLES x 32"}
    # ---- HBM traffic of the whole step
    cf, _ = counters(fetch, ("FETCH_SIZE",))
    cw, _ = counters(write, ("WRITE_SIZE",))
    fb = 2.0 * 1024.0 * sum(v for v, _, _ in cf["FETCH_SIZE"].values())
    wb = 1024.0 * sum(v for v, _, _ in cw["WRITE_SIZE"].values())
    res["hbm"] = {"fetch_gb_per_step": fb / 1e9, "write_gb_per_step": wb / 1e9,
                  "achieved_gb_s_over_step": (fb + wb) / (wall_ms * 1e-3) / 1e9, "peak_gb_s": 8000.0,
                  "frac": (fb + wb) / (wall_ms * 1e-3) / 1e9 / 8000.0,
                  "conv_fwd_k_bytes_per_launch": (2.0 * 1024.0 * sum(v for v, n, _ in cf["FETCH_SIZE"].values() if is_conv(n)) +
                                                  1024.0 * sum(v for v, n, _ in cw["WRITE_SIZE"].values() if is_conv(n))) /
                  max(1, sum(1 for v, n, _ in cf["FETCH_SIZE"].values() if "conv_fwd_k" in n))}
    res["wall_time_used"] = "ms_per_step of the unprofiled bench run" if res["ms_per_step_unprofiled"] else "trace"
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(*sys.argv[1:8])
