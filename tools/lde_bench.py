#!/usr/bin/env python3
"""Per-kernel timing of one Aurora-shaped codeword LDE (2^20 coefficients -> 2^25 points, GF(2^192)) and one full 2^22 FFT."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import libiop_amd
lib = libiop_amd.lib(); lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
out = {}
for name, d, m in (("lde_2^20->2^25", 20, 25), ("fft_2^22", 22, 22)):
    basis = libiop_amd.standard_basis(m)
    shift = np.array([1 << m if m > d else 0, 0, 0], dtype=np.uint64)
    g = torch.Generator(device=dev); g.manual_seed(1)
    c = torch.randint(-2**63, 2**63 - 1, (1 << d, 3), dtype=torch.int64, device=dev, generator=g)
    o = torch.empty((1 << m, 3), dtype=torch.int64, device=dev)
    lib.additive_FFT_dev(c.data_ptr(), 1 << d, basis, shift, o.data_ptr()); lib.synchronize()
    lib.profile_begin()
    reps = 3
    for _ in range(reps):
        lib.additive_FFT_dev(c.data_ptr(), 1 << d, basis, shift, o.data_ptr())
    rep = lib.profile_report()
    out[name] = {"total_ms": round(sum(v[1] for v in rep.values()) / reps, 3), **{k: (v[0] // reps, round(v[1] / reps, 3)) for k, v in rep.items()}}
print(json.dumps(out))
