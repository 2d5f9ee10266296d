#!/usr/bin/env python3
"""Concurrency of the kernel trace (rocprofv3 rocpd database) over its last `frac`: wall span, time with >= 1 and >= 2
kernels in flight, sum of kernel durations, per-queue share.   usage: rocpd_overlap.py results.db [frac=0.4]"""
import sqlite3
import sys

import numpy as np


def main(db, frac=0.4):
    c = sqlite3.connect(db)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like "
                                   "'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch%s)" % suf)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute("select d.start, d.end, %s, s.kernel_name from rocpd_kernel_dispatch%s d join "
                     "rocpd_info_kernel_symbol%s s on d.kernel_id = s.id order by d.start"
                     % (("d." + qcol) if qcol else "0", suf, suf)).fetchall()
    r = rows[int(len(rows) * (1 - frac)):]
    st = np.array([x[0] for x in r], dtype=np.int64)
    en = np.array([x[1] for x in r], dtype=np.int64)
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    depth = np.cumsum(ev[:, 1])[:-1]
    dt = np.diff(ev[:, 0])
    span = en.max() - st.min()
    print("columns:", cols)
    print("kernels %d  span %.2f ms  sum of durations %.2f ms" % (len(r), span / 1e6, (en - st).sum() / 1e6))
    for k in (0, 1, 2, 3):
        sel = depth == k if k < 3 else depth >= 3
        print("  %s kernels in flight: %.2f ms" % (str(k) if k < 3 else ">=3", dt[sel].sum() / 1e6))
    q = {}
    for a, b, qq, _ in r:
        q[qq] = q.get(qq, 0) + (b - a)
    print("per %s:" % qcol, {k: round(v / 1e6, 2) for k, v in q.items()})


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
