#!/usr/bin/env python3
"""Row check, fz and sumcheck-g virtual oracles at BASELINE config-4 scale (2^25-point codeword domain over GF(2^192), 2^20-point
constraint / summation domain), device-resident.  One JSON line: ms per call."""
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
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    h = m - 5
    lib = la.lib()
    lib.init(0)
    dev = torch.device("cuda:0")
    n = 1 << m
    g = torch.Generator(device=dev).manual_seed(1)
    cols = [torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev, generator=g) for _ in range(3)]
    out = torch.empty((n, 3), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    basis, shift = la.standard_basis(m), np.array([1 << m, 0, 0], dtype=np.uint64)
    zero = np.zeros(3, dtype=np.uint64)
    mu = np.array([5, 6, 7], dtype=np.uint64)
    calls = {
        "rowcheck": lambda: lib.rowcheck_dev(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), basis, shift, h, zero, out.data_ptr()),
        "fz": lambda: lib.fz_dev(cols[0].data_ptr(), cols[1].data_ptr(), basis, shift, basis[:4], zero, out.data_ptr()),
        "sumcheck_g": lambda: lib.sumcheck_g_dev(cols[0].data_ptr(), cols[1].data_ptr(), basis, shift, basis[:h], zero, mu, out.data_ptr()),
    }
    res = {"log_n": m, "log_h": h}
    for name, fn in calls.items():
        fn()
        lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        lib.synchronize()
        res[name + "_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
