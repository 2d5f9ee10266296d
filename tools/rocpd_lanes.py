#!/usr/bin/env python3
"""Per-queue kernel timeline of the LAST training step in a rocprofv3 rocpd database (steps are delimited by the
once-per-step loss_pixel_k launch): one line per kernel with start (us from the step's first kernel), duration, queue
and the number of kernels in flight at its start.   usage: rocpd_lanes.py results.db out.txt"""
import re
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like "
                                   "'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch%s)" % suf)]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = c.execute("select d.start, d.end, d.%s, s.kernel_name, d.grid_size_x, d.workgroup_size_x from "
                     "rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id order by d.start"
                     % (qcol, suf, suf)).fetchall()
    marks = [i for i, r in enumerate(rows) if "loss_pixel_k" in r[3]]
    a, b = marks[-3], marks[-2]          # one full step: objective -> backward -> optimizer -> next forward
    step = rows[a:b]
    t0 = step[0][0]
    qs = sorted({r[2] for r in step})
    ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
    wall, depth, prev = [0.0] * 5, 0, ev[0][0]
    for x, k in ev:
        wall[min(depth, 4)] += (x - prev) / 1e3
        prev, depth = x, depth + k
    with open(out, "w") as f:
        f.write("step span %.1f us, %d kernels, queues %s\n" % ((step[-1][1] - t0) / 1e3, len(step), qs))
        f.write("wall time by kernels in flight (us): " + "  ".join("%d%s: %.0f" % (k, "+" if k == 4 else "", wall[k]) for k in range(5))
                + "   sum of kernel durations %.0f\n" % (sum(r[1] - r[0] for r in step) / 1e3))
        for r in step:
            depth = sum(1 for x in step if x[0] <= r[0] < x[1])
            name = re.sub(r"\(.*", "", r[3]).replace("void ", "")
            name = re.sub(r"^_Z\d+", "", name)[:44]
            f.write("%9.1f %7.1f q%-2d x%d %-44s wg %d\n" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, qs.index(r[2]), depth, name,
                                                     r[4] // max(r[5], 1)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
