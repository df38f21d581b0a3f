#!/usr/bin/env python3
"""LDT-reducer combination at BASELINE config-4 scale: 7 oracles over a 2^25-point domain of GF(2^192) (Aurora-like degree
spread: one maximal, six submaximal), device-resident.  Prints one JSON line: ms per call, field products, HBM bytes."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import libiop_amd as la
    import oracle
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    lib = la.lib()
    lib.init(0)
    dev = torch.device("cuda:0")
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << m
    d = m - 5
    degrees = [(1 << (d + 1)) - 1, 1 << d, 1 << d, 1 << d, (1 << d) + 2 * 41 - 1, (1 << (d + 1)) - 2, (1 << d) - 1]
    g = torch.Generator(device=dev).manual_seed(1)
    cols = [torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev, generator=g) for _ in degrees]
    out = torch.empty((n, 3), dtype=torch.int64, device=dev)
    basis, shift = oracle.standard_basis(m, 3), np.array([1 << m, 0, 0], dtype=np.uint64)
    from helpers_bench import rand_words
    coeffs = rand_words(3, 2 * len(degrees))
    run = lambda: lib.ldt_combine_dev([c.data_ptr() for c in cols], degrees, coeffs, basis, shift, out.data_ptr())
    run()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    mx = max(degrees)
    products = n * sum(1 + (bin(mx - dg).count("1") if dg < mx else 0) for dg in degrees)
    print(json.dumps({"workload": "ldt_combine gf192 2^%d x %d oracles" % (m, len(degrees)), "degrees": degrees, "ms": round(ms, 3),
                      "reference_field_products": products,
                      "algorithmic_bytes": (len(degrees) + 1) * n * 24, "GBps": (len(degrees) + 1) * n * 24 / ms / 1e6}))


if __name__ == "__main__":
    main()
