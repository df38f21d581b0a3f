#!/usr/bin/env python3
"""Butterfly-pass rates of ONE polynomial's low-degree extension (2^d coefficients -> 2^m points) over the standard basis (one- and two-word numerators at the
last two levels) and over a general basis (general product at all six levels), per kernel: cycles per wave-butterfly per SIMD at 2.4 GHz, to set
beside the in-register product rates of tools/ubench/mul_rates.  Tuning variables are read from the environment as usual."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import libiop_amd

lib = libiop_amd.lib()
lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
out = {}
for name, d, m, general in (("standard 2^20 -> 2^25", 20, 25, False), ("general 2^19 -> 2^24", 19, 24, True), ("general 2^20 -> 2^25", 20, 25, True)):
    if general:
        rng = np.random.Generator(np.random.PCG64(7))
        basis = rng.integers(0, 2**63, size=(m, 3), dtype=np.uint64)
        shift = rng.integers(0, 2**63, size=3, dtype=np.uint64)
    else:
        basis = libiop_amd.standard_basis(m)
        shift = np.array([1 << m, 0, 0], dtype=np.uint64)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    c = torch.randint(-2**63, 2**63 - 1, (1 << d, 3), dtype=torch.int64, device=dev, generator=g)
    o = torch.empty((1 << m, 3), dtype=torch.int64, device=dev)
    for _ in range(40):                  # half a second of work first: the shader clock ramps up over the first milliseconds of load
        lib.additive_FFT_dev(c.data_ptr(), 1 << d, basis, shift, o.data_ptr())
    lib.synchronize()
    lib.profile_begin()
    reps = 4
    for _ in range(reps):
        lib.additive_FFT_dev(c.data_ptr(), 1 << d, basis, shift, o.data_ptr())
    rep = lib.profile_report()
    products = getattr(lib, "last_profile_products", {})
    row = {}
    for k, v in rep.items():
        if k.startswith("k_bfly") and products.get(k):
            row[k] = {"launches": v[0] // reps, "ms": round(v[1] / reps, 3), "butterflies": products[k] // reps,
                      "cycles_per_wave_butterfly_per_simd": round(v[1] * 1e-3 * 2.4e9 * 1024 / (products[k] / 64), 0)}
    out[name] = row
print(json.dumps(out, indent=1))
