#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into the same columns as `--stats`:
Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs.   usage: rocpd_stats.py results.db out.csv [tail]
tail (default 1.0): only the last fraction of the dispatches (by start time) -- the steady-state iterations, without
the plan-build autotuner's candidate launches."""
import csv
import sqlite3
import sys


def main(db, out, tail=1.0):
    c = sqlite3.connect(db)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like "
                                   "'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    n = c.execute("select count(*) from rocpd_kernel_dispatch%s" % suf).fetchone()[0]
    skip = int(n * (1.0 - tail))
    t0 = c.execute("select start from rocpd_kernel_dispatch%s order by start limit 1 offset %d" % (suf, skip)).fetchone()[0]
    rows = c.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                     "max(d.end-d.start) from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                     "on d.kernel_id = s.id where d.start >= %d group by s.kernel_name order by 3 desc" % (suf, suf, t0)).fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])
    return rows, tot


if __name__ == "__main__":
    rows, tot = main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
    for r in rows[:25]:
        print("%6.2f%%  calls %6d  avg %10.1f us  %s" % (100.0 * r[2] / tot, r[1], r[3] / 1e3, r[0][:110]))
