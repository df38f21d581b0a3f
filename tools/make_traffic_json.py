#!/usr/bin/env python3
"""profiles/r01_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (rocpd databases): per-launch HBM bytes of the
phase-1 kernel with the guide's gfx950 correction (FETCH_SIZE counts half of a coalesced streaming read: doubled; WRITE_SIZE
exact), both in KiB.  Usage: make_traffic_json.py <fetch_dir> <write_dir> <log_n> <out.json> [kernel substring ...]"""
import glob
import hashlib
import json
import os
import sqlite3
import sys
import time


def kernel_sources_digest():
    """The same digest bench.py computes: the figures are only quoted for the kernel sources they were collected on."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for dp, _, fs in sorted(os.walk(os.path.join(root, "libiop_amd", "csrc"))):
        for f in sorted(fs):
            if f.endswith((".hip", ".h")):
                h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()


def avg_counter(d, counter, kernel_like):
    path = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    q = ("select avg(e.value), count(*) from rocpd_pmc_event%s e join rocpd_info_pmc%s p on e.pmc_id = p.id join rocpd_kernel_dispatch%s d "
         "on e.event_id = d.event_id join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id where p.name = ? and s.kernel_name like ?" % (suf, suf, suf, suf))
    return c.execute(q, (counter, kernel_like)).fetchone()


kernels = sys.argv[5:] or ["k_phase1"]
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over bench.py (summaries next to this file in profiles/)",
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read on gfx950 -> doubled; WRITE_SIZE exact; both KiB",
    "log_n": int(sys.argv[3]),
    "kernel_sources_sha256": kernel_sources_digest(),
    "collected": time.strftime("%Y-%m-%d"),
    "kernels": {},
}
for k in kernels:
    fetch, nf = avg_counter(sys.argv[1], "FETCH_SIZE", "%" + k + "%")
    write, nw = avg_counter(sys.argv[2], "WRITE_SIZE", "%" + k + "%")
    if fetch is None or write is None:
        continue
    out["kernels"][k] = {"fetch_size_kib_avg": round(fetch, 1), "write_size_kib_avg": round(write, 1), "launches": nf,
                         "traffic_bytes_per_launch": int(round((2 * fetch + write) * 1024))}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
