#!/usr/bin/env python3
"""tests/golden/transcript_digests.json: SHA-256 of the ORACLE provers' serialized transcripts (and index roots) for small seeded
instances — a regression pin of the oracle itself across rounds (the device provers are compared with the live oracle byte for byte;
this file catches the oracle drifting).  Regenerate only for a deliberate, explained change: python tools/make_transcript_digests.py"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle

CASES = {
    "aurora": [("gf64", oracle.FIELD_GF64, 6, 3, 0x2204), ("gf192", oracle.FIELD_GF192, 6, 3, 0x2204), ("gf192", oracle.FIELD_GF192, 8, 15, 0x2204),
               ("edwards_Fr", oracle.FIELD_EDWARDS, 6, 3, 0x2204), ("edwards_Fr", oracle.FIELD_EDWARDS, 8, 15, 0x2204)],
    "fractal": [("gf64", oracle.FIELD_GF64, 5, 3, 0x2205), ("gf192", oracle.FIELD_GF192, 5, 3, 0x2205), ("edwards_Fr", oracle.FIELD_EDWARDS, 6, 0, 0x2205),
                ("edwards_Fr", oracle.FIELD_EDWARDS, 7, 15, 0x2205)],
    "fri_snark": [("gf192", oracle.FIELD_GF192, 10, 2, 2, 1, 10, 0x2203), ("edwards_Fr", oracle.FIELD_EDWARDS, 10, 2, 2, 1, 10, 0x2203)],
}


def digests():
    out = {"aurora": [], "fractal": [], "fri_snark": []}
    for name, code, log_n, k, seed in CASES["aurora"]:
        t = oracle.aurora_prove(code, log_n, k, seed)
        out["aurora"].append({"field": name, "log_n": log_n, "num_inputs": k, "seed": seed, "bytes": len(t), "sha256": hashlib.sha256(t).hexdigest()})
    for name, code, log_n, k, seed in CASES["fractal"]:
        t, roots = oracle.fractal_prove(code, log_n, k, seed)
        out["fractal"].append({"field": name, "log_n": log_n, "num_inputs": k, "seed": seed, "bytes": len(t), "sha256": hashlib.sha256(t).hexdigest(),
                               "index_roots": [r.hex() for r in roots]})
    for name, code, dim, rs, loc, inter, queries, seed in CASES["fri_snark"]:
        t = oracle.fri_snark_prove(code, dim, rs, loc, inter, queries, seed)
        out["fri_snark"].append({"field": name, "dim": dim, "rs_extra": rs, "localization": loc, "interactions": inter, "queries": queries, "seed": seed,
                                 "bytes": len(t), "sha256": hashlib.sha256(t).hexdigest()})
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "transcript_digests.json")
    json.dump(digests(), open(path, "w"), indent=1)
    print("wrote", path)
