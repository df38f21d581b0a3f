#!/usr/bin/env python3
"""Which kernels do the constant-carrying launches (k_upload_small) of one proof period feed?  From a rocprofv3 --kernel-trace database: for every run
of consecutive k_upload_small launches, the kernel that follows it.  Usage: upload_consumers.py <results dir or .db> [--period-kernel k_lincheck_add]"""
import argparse, collections, glob, sqlite3
ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--period-kernel", default="k_lincheck_add")
ap.add_argument("--period-index", type=int, default=3)
a = ap.parse_args()
path = a.path if a.path.endswith(".db") else glob.glob(a.path + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id order by d.start" % (suf, suf)).fetchall()
marks = [r[0] for r in rows if a.period_kernel in r[2]]
rows = [r for r in rows if marks[a.period_index] <= r[0] < marks[a.period_index + 1]]
hist = collections.Counter()
run = 0
for r in rows:
    if "k_upload_small" in r[2]:
        run += 1
        continue
    if run:
        hist[r[2].split("iopx")[-1][:44]] += run
    run = 0
for k, v in hist.most_common():
    print("%4d  %s" % (v, k))
print("total", sum(hist.values()))
