#!/usr/bin/env python3
"""One-off record of the CPU baseline at the headline sizes: the oracle's literal restatement of the reference's Aurora
prover (oracle/aurora.hpp; PCLMUL gf192, one thread, as the reference is single-threaded) timed at 2^k constraints for the given k — same
protocol parameters and seed as bench.py's workload.  Writes a JSON with seconds, the reference-count field-ops/s (bench.py's numerator) and
the host it ran on; bench.py's cpu_baseline stays the bounded 2^13 sample, this file is what "vs CPU at 2^20" is read from.

    python tools/cpu_baseline_sizes.py --log-n 16 18 20 --out profiles/r04_cpu_baseline_sizes.json
Test infrastructure is used here as the measured baseline only (kind "port")."""
import argparse
import hashlib
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, nargs="+", default=[16, 18])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    import oracle
    from libiop_amd import aurora, domains
    res = {"what": "oracle.aurora_prove (CPU restatement of the reference prover, one thread), GF(2^192), generate_r1cs_example(n, 15, n - 1), seed 0x%x, "
                   "security 128, RS_extra_dimensions 5, localization 2" % bench.SEED,
           "kind": "port", "cores": 1, "host": bench.host_description(), "sizes": []}
    out = a.out
    for k in a.log_n:
        t0 = time.perf_counter()
        tr = oracle.aurora_prove(oracle.FIELD_GF192, k, 15, bench.SEED)
        s = time.perf_counter() - t0
        p = aurora.AuroraParameters(domains.GF192(), 1 << k, (1 << k) - 1, 15)
        inv = bench.aurora_transform_inventory(k, p.RS_extra_dimensions, p.codeword_domain_dim - sum(p.localization_parameters))
        ops = sum(sum(bench.ref_fft_ops(m)) for _, m in inv)
        res["sizes"].append({"log_n": k, "prover_seconds": s, "ref_fft_field_ops": ops, "field_ops_per_s": ops / s, "argument_bytes": len(tr),
                             "transcript_blake2b": hashlib.blake2b(tr, digest_size=32).hexdigest(),
                             "peak_rss_gib": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20})
        print(json.dumps(res["sizes"][-1]), flush=True)
        if out:
            os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
            with open(out, "w") as f:
                json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
