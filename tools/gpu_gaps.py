#!/usr/bin/env python3
"""GPU idle gaps inside the last proof of a rocprofv3 --kernel-trace run (rocpd database): every pause longer than --min-us between
the end of one kernel and the start of the next, with the kernels on both sides — where host work or a host read-back is exposed.
Usage: gpu_gaps.py <results dir or .db> [--min-us 100] [--last-ms 120]"""
import argparse, glob, sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--min-us", type=float, default=100.0)
ap.add_argument("--context", type=int, default=0, help="also print this many kernels before and after each listed gap")
ap.add_argument("--last-ms", type=float, default=120.0, help="look at the kernels of the last this-many milliseconds of the trace")
ap.add_argument("--end-offset-ms", type=float, default=0.0, help="the window ends this many milliseconds before the last kernel (skip a tail that ran under other instrumentation)")
ap.add_argument("--period-kernel", default=None, help="window = from the start of the I-th launch of this kernel (substring) to the start of the next one: for a kernel "
                "that runs once per proof, exactly one proof period whatever the tracer's slowdown")
ap.add_argument("--period-index", type=int, default=3)
ap.add_argument("--histogram", action="store_true", help="idle time by gap length, and by the kernel the GPU waited for (where the launch-bound time sits)")
a = ap.parse_args()
path = a.path if a.path.endswith(".db") else glob.glob(a.path + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id order by d.start" % (suf, suf)).fetchall()
if a.period_kernel:
    marks = [r[0] for r in rows if a.period_kernel in r[2]]
    lo_t, hi_t = marks[a.period_index], marks[a.period_index + 1]
    rows = [r for r in rows if lo_t <= r[0] < hi_t]
else:
    t_end = rows[-1][1] - a.end_offset_ms * 1e6
    rows = [r for r in rows if t_end - a.last_ms * 1e6 <= r[0] and r[1] <= t_end]
busy = sum(r[1] - r[0] for r in rows)
span = rows[-1][1] - rows[0][0]
short = lambda n: n.split("iopx")[-1][:40]
gaps = [(rows[i + 1][0] - rows[i][1], short(rows[i][2]), short(rows[i + 1][2]), i) for i in range(len(rows) - 1)]
print("window %.2f ms, kernels busy %.2f ms, idle %.2f ms in %d gaps (%d >= %.0f us)" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps),
      sum(1 for g in gaps if g[0] >= a.min_us * 1e3), a.min_us))
if a.histogram:
    buckets = [(0, 3), (3, 10), (10, 30), (30, 100), (100, 1e9)]
    for lo, hi in buckets:
        sel = [g for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
        print("  gaps of %5.0f .. %-8s us: %5d, %.3f ms" % (lo, "%.0f" % hi if hi < 1e8 else "inf", len(sel), sum(g[0] for g in sel) / 1e6))
    by_next = {}
    for g in gaps:
        k = g[2].split("E")[0][:36]
        n, t = by_next.get(k, (0, 0))
        by_next[k] = (n + 1, t + g[0])
    for k, (n, t) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  idle before %-38s %5d gaps, %.3f ms" % (k, n, t / 1e6))
for g, before, after, i in sorted(gaps, reverse=True):
    if g < a.min_us * 1e3:
        break
    print("%8.1f us  after %-42s before %s" % (g / 1e3, before, after))
    if a.context:
        lo, hi = max(0, i - a.context + 1), min(len(rows), i + 1 + a.context)
        print("             " + " | ".join("%s (%.0f us)" % (short(r[2])[:28], (r[1] - r[0]) / 1e3) for r in rows[lo:i + 1]) + "  >>>  " +
              " | ".join("%s (%.0f us)" % (short(r[2])[:28], (r[1] - r[0]) / 1e3) for r in rows[i + 1:hi]))
