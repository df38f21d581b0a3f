#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel launch statistics and, when the run collected a
PMC counter, the per-kernel average of that counter.  Usage: rocprof_summary.py <results.db> [...]"""
import sqlite3
import sys


def summarise(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    kd, ks, pe, pi = ("rocpd_kernel_dispatch" + suf, "rocpd_info_kernel_symbol" + suf, "rocpd_pmc_event" + suf, "rocpd_info_pmc" + suf)
    print("# %s" % path)
    print("%-58s %8s %12s %12s %12s %12s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us"))
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
         "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    for r in c.execute(q):
        print("%-58s %8d %12.3f %12.3f %12.3f %12.3f" % (r[0][:58], r[1], r[2], r[3], r[4], r[5]))
    if c.execute("select count(*) from %s" % pe).fetchone()[0]:
        print("%-58s %-14s %8s %16s %16s" % ("kernel", "counter", "calls", "avg_value", "sum_value"))
        q = ("select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value) from %s e join %s p on e.pmc_id = p.id "
             "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, p.name order by 5 desc" % (pe, pi, kd, ks))
        for r in c.execute(q):
            print("%-58s %-14s %8d %16.1f %16.1f" % (r[0][:58], r[1], r[2], r[3], r[4]))
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
