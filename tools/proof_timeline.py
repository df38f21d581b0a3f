#!/usr/bin/env python3
"""The kernels of the last proof of a rocprofv3 --kernel-trace run (rocpd database) in launch order, runs of the same kernel folded:
name, launches, total us, gap before the run.  Usage: proof_timeline.py <results dir or .db> [--last-ms 67]"""
import argparse, glob, sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--last-ms", type=float, default=67.0)
a = ap.parse_args()
path = a.path if a.path.endswith(".db") else glob.glob(a.path + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id order by d.start" % (suf, suf)).fetchall()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - a.last_ms * 1e6]
short = lambda n: n.split("iopx")[-1].split("E")[0][:34] if "iopx" in n else n[:34]
runs = []
for i, (s, e, n) in enumerate(rows):
    gap = (s - rows[i - 1][1]) / 1e3 if i else 0.0
    if runs and runs[-1][0] == short(n):
        runs[-1][1] += 1; runs[-1][2] += (e - s) / 1e3; runs[-1][3] += gap
    else:
        runs.append([short(n), 1, (e - s) / 1e3, gap, (s - rows[0][0]) / 1e6])
for name, cnt, us, gap, at in runs:
    print("%8.2f ms  %-36s x%-3d %9.1f us   gaps %7.1f us" % (at, name, cnt, us, gap))
