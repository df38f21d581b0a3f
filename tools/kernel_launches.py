#!/usr/bin/env python3
"""Per-launch durations (microseconds, launch order) of the kernels whose name contains a pattern, from a rocprofv3 rocpd
database directory.  Usage: kernel_launches.py <dir> <pattern> [last_n]"""
import glob
import sqlite3
import sys

for path in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    q = ("select s.kernel_name, d.grid_size_x, (d.end-d.start)/1e3 from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
         "on d.kernel_id = s.id where s.kernel_name like ? order by d.start" % (suf, suf))
    rows = list(c.execute(q, ("%" + sys.argv[2] + "%",)))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
    for r in rows[-n:]:
        print(r[0][:40], r[1], round(r[2], 1))
