#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (rocpd database) plus the derived fractions the guide's
identity gives (MI355X_MICROARCH.md, rocprofv3 PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).
Usage: make_sq_json.py <pmc_dir> <out.json> <kernel substring> ..."""
import glob
import json
import sqlite3
import sys

path = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
out = {"source": "rocprofv3 --pmc (one pass, SQ block) over the command in the file name; averages per launch", "kernels": {}}
for k in sys.argv[3:]:
    q = ("select p.name, avg(e.value), count(*) from rocpd_pmc_event%s e join rocpd_info_pmc%s p on e.pmc_id = p.id join rocpd_kernel_dispatch%s d "
         "on e.event_id = d.event_id join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id where s.kernel_name like ? group by p.name" % (suf, suf, suf, suf))
    rows = c.execute(q, ("%" + k + "%",)).fetchall()
    if not rows:
        continue
    v = {name: val for name, val, _ in rows}
    d = {"launches": rows[0][2], "counters": {n: round(x, 1) for n, x in v.items()}}
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        for n, label in (("SQ_ACTIVE_INST_VALU", "valu_active_frac"), ("SQ_ACTIVE_INST_ANY", "any_active_frac"), ("SQ_WAIT_ANY", "parked_frac"),
                         ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_LDS", "lds_active_frac")):
            if n in v:
                d[label] = round(v[n] / wc, 4)
    out["kernels"][k] = d
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
