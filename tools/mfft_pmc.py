#!/usr/bin/env python3
"""Per-launch PMC values of k_mfft_pass from a rocprofv3 rocpd database (launch order), to compare the passes of one FFT."""
import glob
import sqlite3
import sys

for path in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    q = ("select d.start, p.name, e.value, (d.end-d.start)/1e3 from rocpd_pmc_event%s e join rocpd_info_pmc%s p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch%s d on e.event_id = d.event_id join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id "
         "where s.kernel_name like '%%mfft%%' order by d.start" % (suf, suf, suf, suf))
    rows = {}
    for start, name, value, dur in c.execute(q):
        rows.setdefault(start, {"us": dur})[name] = rows.get(start, {}).get(name, 0) + value
    keys = sorted(rows)
    for k in keys[6:15]:
        print({n: round(v, 1) for n, v in rows[k].items()})
