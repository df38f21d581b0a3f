#!/usr/bin/env python3
"""Digests of the oracle's Fractal index roots and transcript at sizes the oracle needs minutes for (181-bit field, k = 0, seed 0x2205,
RS_extra_dimensions 3, localization 2: BASELINE configs[4]'s parameters), for tests/golden/oracle_fractal_transcript_digests_large.json.
    python tools/cpu_digest_fractal.py --log-n 16 --out gpurun_out/r04_fractal_digests.json
The oracle holds every codeword of the proof on the host (about 0.8 GB at 2^12, growing linearly): 2^16 needs about 13 GB."""
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
    ap.add_argument("--log-n", type=int, nargs="+", default=[16])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import oracle
    res = {"what": "oracle.fractal_prove(FIELD_EDWARDS, log_n, 0, 0x2205): BLAKE2b-256 of the transcript bytes, the index Merkle roots", "digests": {}}
    for k in a.log_n:
        t0 = time.perf_counter()
        t, roots = oracle.fractal_prove(oracle.FIELD_EDWARDS, k, 0, 0x2205)
        res["digests"][str(k)] = {"transcript_blake2b": hashlib.blake2b(t, digest_size=32).hexdigest(), "argument_bytes": len(t),
                                  "index_roots": [bytes(r).hex() for r in roots], "oracle_seconds": time.perf_counter() - t0,
                                  "peak_rss_gib": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20}
        print(json.dumps(res["digests"][str(k)]), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
