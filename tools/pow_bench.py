#!/usr/bin/env python3
"""Proof-of-work grind at the difficulty of BASELINE config 4 (pow_bitlen = dim_h + 3 = 23, common_bcs_parameters.tcc:23-25):
wall time of iopx_pow_solve_* over a few challenges, and the oracle (one CPU core) on an easier instance scaled up."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import libiop_amd as la
    import oracle
    lib = la.lib()
    lib.init(0)
    bitlen = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    rng = np.random.default_rng(1)
    out = {"pow_bitlen": bitlen}
    lib.solve_pow(bytes(32), 4)
    ts, tries = [], []
    for _ in range(8):
        ch = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
        t0 = time.perf_counter()
        ans = lib.solve_pow(ch, bitlen)
        ts.append(time.perf_counter() - t0)
        assert oracle.pow_verify_blake2b(ch, ans, bitlen)
        tries.append(1 if ans == ch else int.from_bytes(ans[24:], "little") + 2)
    out["blake2b"] = {"ms": [round(t * 1e3, 3) for t in ts], "candidates": tries, "hashes_per_s_incl_launch": sum(tries) / sum(ts)}
    t0 = time.perf_counter()
    _, calls = oracle.pow_solve_blake2b(bytes(rng.integers(0, 256, size=32, dtype=np.uint8)), 18)
    out["blake2b_cpu_hashes_per_s_1core"] = calls / (time.perf_counter() - t0)
    with open(os.path.join(ROOT, "libiop_amd", "data", "poseidon_alt_bn128.json")) as f:
        sets = json.load(f)["sets"]
    for name in ["starkware_alpha5_t3", "high_alpha17_t3"]:
        p, po = la.PoseidonParams.shipped(name), oracle.PoseidonParams(sets[name])
        lib.solve_pow(np.zeros(4, dtype=np.uint64), 2, poseidon_params=p)
        ts, tries = [], []
        for i in range(4):
            ch = oracle.bn_from_ints([int.from_bytes(rng.bytes(31), "little")])[0]
            t0 = time.perf_counter()
            ans = lib.solve_pow(ch, bitlen, poseidon_params=p)
            ts.append(time.perf_counter() - t0)
            assert oracle.pow_verify_poseidon(po, ch, ans, bitlen)
            tries.append(oracle.bn_to_ints(ans[None, :])[0] + 1)
        t0 = time.perf_counter()
        _, calls = oracle.pow_solve_poseidon(po, oracle.bn_from_ints([12345])[0], 12)
        out[name] = {"ms": [round(t * 1e3, 3) for t in ts], "candidates": tries, "hashes_per_s_incl_launch": sum(tries) / sum(ts),
                     "cpu_hashes_per_s_1core": calls / (time.perf_counter() - t0)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
