#!/usr/bin/env python3
"""Throughput of the Poseidon Merkle kernels (alt_bn128 Fr): one tree over `--oracles` columns of 2^log_n elements with
cosets of `--coset`, for every shipped parameter set.  Prints one JSON line per set: ms per tree, permutations / s
(leaf chunks + inner nodes), per-kernel HIP-event times.  --cpu times the oracle's permutation on one core beside it."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=21)
    ap.add_argument("--oracles", type=int, default=1)
    ap.add_argument("--coset", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    import torch
    import libiop_amd as la
    lib = la.lib()
    lib.init(0)
    dev = torch.device("cuda:0")
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    n, cs, r = 1 << args.log_n, args.coset, args.oracles
    L = n // cs
    rng = np.random.Generator(np.random.PCG64(5))
    raw = rng.integers(0, 2**62, size=(r, n, 4), dtype=np.uint64)
    raw[..., 3] >>= 10                                  # below p: valid Montgomery representatives
    cols = [torch.from_numpy(raw[k].view(np.int64)).to(dev) for k in range(r)]
    nodes = torch.empty((2 * L - 1, 4), dtype=torch.int64, device=dev)
    for name in ["starkware_alpha5_t3", "high_alpha17_t3", "high_alpha17_t4"]:
        p = la.PoseidonParams.shipped(name)
        run = lambda: lib.merkle_tree_poseidon_dev(p, [c.data_ptr() for c in cols], n, cs, nodes.data_ptr())
        run()
        torch.cuda.synchronize()
        lib.profile_begin()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.reps
        prof = lib.profile_report()
        chunks = -(-(r * cs) // p.rate)
        perms = L * chunks + (L - 1)
        out = {"set": name, "log_n": args.log_n, "oracles": r, "coset": cs, "ms_per_tree": round(ms, 3), "permutations": perms,
               "perm_per_s": perms / ms * 1e3, "kernels_ms": {k: round(v[1] / args.reps, 3) for k, v in prof.items()},
               "root": np.frombuffer(lib.read_digest(nodes.data_ptr()), dtype=np.uint64).tolist()}
        if args.cpu:
            import oracle
            with open(os.path.join(ROOT, "libiop_amd", "data", "poseidon_alt_bn128.json")) as f:
                po = oracle.PoseidonParams(json.load(f)["sets"][name])
            st = np.zeros((p.state_size, 4), dtype=np.uint64)
            t0 = time.perf_counter()
            cnt = 0
            while time.perf_counter() - t0 < 2.0:
                for _ in range(200):
                    st = oracle.poseidon_permute(po, st)
                cnt += 200
            out["cpu_perm_per_s_1core"] = cnt / (time.perf_counter() - t0)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
