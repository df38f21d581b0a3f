#!/usr/bin/env python3
"""tests/golden/oracle_general_r1cs_digests.json: BLAKE2b-256 of the ORACLE provers' transcripts (and index roots) on general constraint systems
(tests/r1cs_general.py: the shapes r1cs_constraint_system::add_constraint builds) at sizes the oracle needs minutes for, so that the GPU tests can
compare the device provers' bytes with them in seconds (tests/test_gpu_general_r1cs.py).  Run on the CPU: python tools/make_general_digests.py"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import r1cs_general as rg

CASES = [("aurora", "gf192", 1 << 16, 15, 92, None), ("aurora", "edwards_Fr", 1 << 16, 15, 93, None),
         ("fractal", "edwards_Fr", 1 << 15, 15, 94, 1 << 15), ("fractal", "gf192", 1 << 10, 15, 95, 1 << 10)]
CODES = {"gf192": oracle.FIELD_GF192, "edwards_Fr": oracle.FIELD_EDWARDS}

out = {"what": "oracle.aurora_prove_csr / fractal_prove_csr on rg.generate(field, n, n - 1, num_inputs, seed, max_nnz): BLAKE2b-256 of the transcript bytes, the index roots",
       "generator": "tools/make_general_digests.py", "cases": []}
for protocol, field, n, k, seed, max_nnz in CASES:
    inst = rg.generate(field, n, n - 1, k, seed, max_nnz=max_nnz)
    args = (CODES[field], inst.matrices, inst.num_variables, inst.num_inputs)
    assert oracle.r1cs_check_csr(*args, inst.assignment)[0] == 0
    t0 = time.perf_counter()
    if protocol == "aurora":
        t, roots = oracle.aurora_prove_csr(*args, inst.assignment), []
    else:
        t, roots = oracle.fractal_prove_csr(*args, inst.assignment)
    entry = {"protocol": protocol, "field": field, "num_constraints": n, "num_inputs": k, "seed": seed, "max_nnz": max_nnz, "nnz": inst.nnz(),
             "transcript_blake2b": hashlib.blake2b(t, digest_size=32).hexdigest(), "argument_bytes": len(t), "index_roots": [r.hex() for r in roots],
             "oracle_seconds": round(time.perf_counter() - t0, 1)}
    print(json.dumps(entry), flush=True)
    out["cases"].append(entry)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_general_r1cs_digests.json"), "w"), indent=1)
