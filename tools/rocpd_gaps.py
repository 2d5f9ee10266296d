#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 rocpd database: busy vs span over the last `frac` of the trace.
usage: rocpd_gaps.py results.db [frac=0.5]"""
import sqlite3, sys
import numpy as np

def main(db, frac=0.5):
    c = sqlite3.connect(db)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like "
                                   "'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                     "on d.kernel_id = s.id order by d.start" % (suf, suf)).fetchall()
    n = len(rows); k0 = int(n * (1 - frac))
    r = rows[k0:]
    st = np.array([x[0] for x in r], dtype=np.int64); en = np.array([x[1] for x in r], dtype=np.int64)
    span = en.max() - st.min(); busy = (en - st).sum()
    gaps = st[1:] - np.maximum.accumulate(en)[:-1]
    pos = gaps[gaps > 0]
    print("kernels %d  span %.2f ms  busy %.2f ms (%.1f%%)  idle %.2f ms" % (len(r), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
    print("gaps: n %d  median %.2f us  mean %.2f us  p90 %.2f us  >20us: %d (%.2f ms)" % (
        len(pos), np.median(pos) / 1e3, pos.mean() / 1e3, np.percentile(pos, 90) / 1e3, (pos > 20000).sum(), pos[pos > 20000].sum() / 1e6))
    big = np.argsort(-gaps)[:12]
    for i in sorted(big):
        print("  gap %8.1f us after %-60s before %s" % (gaps[i] / 1e3, r[i][2][:60], r[i + 1][2][:60]))

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
