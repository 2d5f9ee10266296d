#!/usr/bin/env python3
"""Average PMC counter values per kernel name from a rocprofv3 rocpd database. usage: pmc_summary.py results.db [filter]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def tab(prefix):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like '%s%%'" % prefix)][0]
kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
cols = [r[1] for r in c.execute("pragma table_info(%s)" % pe)]
q = ("select s.kernel_name, p.name, avg(e.value), count(*), avg(d.end-d.start) from %s e join %s p on e.pmc_id=p.id "
     "join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, p.name" % (pe, pi, kd, ks))
res = collections.defaultdict(dict)
for kn, pn, v, n, dur in c.execute(q):
    if flt in kn:
        res[kn][pn] = v
        res[kn]["_n"] = n
        res[kn]["_us"] = dur / 1e3
for kn, d in res.items():
    print(kn[:90], "calls", d.pop("_n"), "avg_us %.1f" % d.pop("_us"))
    for k in sorted(d):
        print("    %-28s %.4g" % (k, d[k]))
