#!/usr/bin/env python3
"""Per-op kernel durations of one graph-replayed training step: joins a rocprofv3 rocpd kernel trace with the op
list `bench.py --profile-out` writes (same process), by walking both in launch order.

usage: rocpd_join.py results.db ops.txt step_index [out.txt]
  step_index = which step (0-based count of pack_k launch groups) to analyse (warmup + steps - 1 = last timed step).
  Run the traced process with PMF_LANES=0: the join walks kernels in start order, which is the op order only on one lane."""
import sqlite3, sys


def klass(name):
    if "conv_fwd_k" in name or "conv_ws_k" in name: return "C"
    if "conv_finish_k" in name: return "F"
    if "wgrad_reduce" in name: return "R"
    if "conv_wgrad" in name or "wgrad_fewc" in name or "wgrad_1x1" in name or "wgrad_stream" in name: return "W"
    return None


def main(db, ops_file, step, out=None):
    c = sqlite3.connect(db)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like "
                                   "'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    rows = c.execute("select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                     "from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                     "on d.kernel_id = s.id order by d.start" % (suf, suf)).fetchall()
    packs = [i for i, r in enumerate(rows) if "pack_k" in r[2] and (i == 0 or "pack_k" not in rows[i - 1][2])]
    lo, hi = packs[step], packs[step + 1]
    ker = rows[lo:hi]
    span = (max(r[1] for r in ker) - ker[0][0]) / 1e3
    busy = sum(r[1] - r[0] for r in ker) / 1e3
    ops = []
    for line in open(ops_file):
        p = line.split()
        if p[1] in ("OP_CONV", "OP_WGRAD_PART", "OP_WGRAD_RED", "OP_WGRAD_RED_MULTI", "OP_WGRAD"):
            i = line.index("[") if "[" in line else -1
            shape = line[i:line.index("]") + 1] if i >= 0 else ""
            gf = float(p[-4]) if i >= 0 else 0.0
            ops.append([p[0], p[1], p[2], p[3] if i >= 0 else "", shape, gf])
    cls = [(klass(r[2]), r) for r in ker]
    cls = [x for x in cls if x[0]]
    k = 0
    res = []
    for op in ops:
        want = {"OP_CONV": "C", "OP_WGRAD_PART": "W", "OP_WGRAD": "W", "OP_WGRAD_RED": "R", "OP_WGRAD_RED_MULTI": "R"}[op[1]]
        if k >= len(cls):
            break
        if op[1] in ("OP_WGRAD_RED", "OP_WGRAD_RED_MULTI"):
            if cls[k][0] == "R":
                r = cls[k][1]; k += 1
                res.append((op, (r[1] - r[0]) / 1e3, 0.0, r))
            continue
        if cls[k][0] != want:
            print("misaligned at op", op, "kernel", cls[k][1][2][:50]); break
        r = cls[k][1]; k += 1
        t, tf = (r[1] - r[0]) / 1e3, 0.0
        if want == "C" and k < len(cls) and cls[k][0] == "F":
            tf = (cls[k][1][1] - cls[k][1][0]) / 1e3; k += 1
        res.append((op, t, tf, r))
    f = open(out, "w") if out else sys.stdout
    f.write("step span %.1f us, kernel busy %.1f us, %d kernels; conv-class kernels joined: %d of %d\n" % (
        span, busy, len(ker), k, len(cls)))
    for op, t, tf, r in res:
        name = r[2]
        a = name.find("ILi")
        tmpl = name[a:name.find("EE", a)].replace("ILi", "").replace("ELi", ",") if a >= 0 else ""
        tot = t + tf
        f.write("%s %-13s %-12s %-20s %-34s %8.1f us +fin %6.1f  %7.2f GF %7.2f TF/s  <%s> grid %dx%dx%d\n" % (
            op[0], op[1], op[2], op[3], op[4], t, tf, op[5], op[5] / tot * 1e3 if tot else 0, tmpl,
            r[3] // max(r[6], 1), r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
