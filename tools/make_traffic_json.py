#!/usr/bin/env python3
"""profiles/rNN_traffic_*.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (rocpd databases) of one command: HBM bytes per launch of
EVERY kernel symbol that takes at least --min-ms of the run (default 0.2 ms per profiled step), one entry per symbol (template instantiations
are kept apart: k_bfly_edge<false, false>, k_bfly_edge<true, false> and k_bfly_edge_fwd_batch are three kernels), with the guide's gfx950
correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half of a coalesced streaming read: doubled; WRITE_SIZE exact; both KiB) and,
beside each, the kernel's ALGORITHMIC bytes per launch as the library's own profiler reports them in the bench / tool JSON of the same
command (--bench-json: `roofline.kernel_algorithmic_bytes_per_launch` of bench.py, or `runs[-1].kernels` of tools/fractal_bench.py).

Usage: make_traffic_json.py <fetch_dir> <write_dir> <log_n> <out.json> [--bench-json FILE] [--min-ms X] [--steps N]
The JSON is keyed by the library's profile name of the symbol (k_bfly_upper_fwd = k_bfly_upper<false, ...>), which is what bench.py looks up."""
import argparse
import glob
import hashlib
import json
import os
import re
import sqlite3
import time


def kernel_sources_digest():
    """The same digest bench.py computes: the figures are only quoted for the kernel sources they were collected on."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for sub in ("csrc", "cpp"):
        for dp, _, fs in sorted(os.walk(os.path.join(root, "libiop_amd", sub))):
            for f in sorted(fs):
                if f.endswith((".hip", ".h", ".hpp")):
                    h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()


def per_symbol(d, counter):
    """{kernel symbol: (average counter value per launch, launches, total ms)}"""
    path = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    q = ("select s.kernel_name, avg(e.value), count(*), sum(d.end - d.start) / 1e6 from rocpd_pmc_event%s e join rocpd_info_pmc%s p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch%s d on e.event_id = d.event_id join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id where p.name = ? "
         "group by s.kernel_name" % (suf, suf, suf, suf))
    return {r[0]: (r[1], r[2], r[3]) for r in c.execute(q, (counter,))}


def profile_name(symbol):
    """The library's profile name (runtime.h ProfScope) of a kernel symbol — mangled, as rocprofv3's database holds it, or demangled: the
    bare kernel name, with _fwd / _inv for the transform kernels whose first template argument is the direction."""
    sym = symbol.strip()
    name, first = None, None
    if sym.startswith("_ZN"):                                 # nested name: <len><identifier> ..., the last one is the kernel
        pos, ident = 3, None
        while pos < len(sym) and sym[pos].isdigit():
            m = re.match(r"\d+", sym[pos:])
            n = int(m.group(0))
            ident = sym[pos + len(m.group(0)):pos + len(m.group(0)) + n]
            pos += len(m.group(0)) + n
        if ident:
            name = ident
            t = re.match(r"ILb([01])E", sym[pos:])
            first = t.group(1) if t else None
    if name is None:
        m = re.match(r"(?:void\s+)?(?:iopx::)?(?:\(anonymous namespace\)::)?(\w+)(?:<(.*)>)?", sym)
        name = m.group(1)
        targ = (m.group(2) or "").split(",")[0].strip()
        first = "1" if targ in ("true", "1", "(bool)1") else ("0" if targ in ("false", "0", "(bool)0") else None)
    if name == "k_bfly_edge_multi":                                            # the edge pass with several cosets per workgroup: the library's profiler names it like k_bfly_edge
        name = "k_bfly_edge"
    if name in ("k_bfly_upper", "k_bfly_edge", "k_phase1"):
        return name + ("_inv" if first == "1" else "_fwd")
    if name == "k_merkle_leaves_sub24":                                        # template arguments: oracles, coset size (, position map)
        m = re.search(r"k_merkle_leaves_sub24ILi(\d+)ELi(\d+)E", sym) or re.search(r"k_merkle_leaves_sub24<(\d+),\s*(\d+)", sym)
        if m:
            return "k_merkle_leaves_%sx%s" % (m.group(1), m.group(2))
    if name in ("k_fri_fold_fused", "k_fri_fold_fused_mul"):                  # template argument: the localization parameter
        m = re.search(r"%sILi(\d+)E" % name, sym) or re.search(r"%s<(\d+)>" % name, sym)
        if m:
            return "%s_eta%s" % (name, m.group(1))
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("log_n", type=int)
    ap.add_argument("out")
    ap.add_argument("--bench-json", default=None)
    ap.add_argument("--min-ms", type=float, default=0.2)
    ap.add_argument("--steps", type=int, default=None, help="proofs the profiled command ran (warm-up included): per-proof figures = totals / steps")
    a = ap.parse_args()
    fetch, write = per_symbol(a.fetch_dir, "FETCH_SIZE"), per_symbol(a.write_dir, "WRITE_SIZE")
    alg = {}
    if a.bench_json:
        text = open(a.bench_json).read().strip()
        j = json.loads(text.splitlines()[-1]) if not text.startswith("{\n") else json.loads(text)
        if "roofline" in j:
            alg = dict(j["roofline"].get("kernel_algorithmic_bytes_per_launch", {}))
        elif "runs" in j:
            alg = {k: v["bytes"] / v["launches"] for k, v in j["runs"][-1].get("kernels", {}).items() if v.get("bytes")}
    steps = a.steps or 1
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over the command (tools/collect_profiles.sh); one entry per kernel SYMBOL",
        "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read on gfx950 -> doubled; WRITE_SIZE exact; both KiB",
        "log_n": a.log_n, "kernel_sources_sha256": kernel_sources_digest(), "collected": time.strftime("%Y-%m-%d"),
        "steps_in_profiled_command": steps, "kernels": {},
    }
    # the algorithmic bytes are per PROFILE NAME: where several symbols share one (k_bfly_edge_multi for the large shapes and k_bfly_edge for the
    # small ones are both k_bfly_edge_fwd), the ratio is taken over all of the name's launches, the short ones below --min-ms included
    by_name = {}
    for sym, (f_avg, n, ms) in fetch.items():
        if sym in write:
            t = by_name.setdefault(profile_name(sym), [0.0, 0, 0])
            t[0] += (2 * f_avg + write[sym][0]) * 1024 * n
            t[1] += n
            t[2] += 1
    for sym, (f_avg, n, ms) in sorted(fetch.items(), key=lambda kv: -kv[1][2]):
        if sym not in write or ms / steps < a.min_ms:
            continue
        name = profile_name(sym)
        traffic = int(round((2 * f_avg + write[sym][0]) * 1024))
        entry = {"symbol": sym, "launches": n, "launches_per_step": n / steps, "ms_per_step_under_pmc": round(ms / steps, 3),
                 "fetch_size_kib_avg": round(f_avg, 1), "write_size_kib_avg": round(write[sym][0], 1), "traffic_bytes_per_launch": traffic}
        if name in alg:
            entry["algorithmic_bytes_per_launch"] = int(round(alg[name]))
            tot, launches, nsym = by_name[name]
            entry["traffic_over_algorithmic"] = round(tot / launches / alg[name], 3)
            if nsym > 1:
                entry["profile_name_launches_per_step"] = launches / steps
                entry["note"] = "%d symbols share this profile name: the ratio is over all %d launches of the name (average %d bytes per launch)" % (nsym, launches, round(tot / launches))
        if name in out["kernels"]:          # two symbols with one profile name (comb / general instantiations): keep both, the busier one under the plain key
            name = name + " [" + sym + "]"
        out["kernels"][name] = entry
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: (v["traffic_bytes_per_launch"], v.get("traffic_over_algorithmic")) for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main()
