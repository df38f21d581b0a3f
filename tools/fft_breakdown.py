#!/usr/bin/env python3
"""Per-kernel time of one standalone additive FFT (BASELINE configs[1]: 2^22 coefficients over the standard basis, shift 0) on cuda:0,
with IOPX_PROFILE_LEVELS=1 the phase-1 passes by level.  Usage: fft_breakdown.py [log_n]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libiop_amd
from libiop_amd import domains

m = int(sys.argv[1]) if len(sys.argv) > 1 else 22
lib = libiop_amd.Library()
lib.set_stream(torch.cuda.current_stream().cuda_stream)
ops = domains.DeviceOps(lib, torch, torch.device("cuda:0"), domains.GF192())
basis, shift = libiop_amd.standard_basis(m), np.zeros(3, dtype=np.uint64)
d_in = ops.upload(np.random.Generator(np.random.PCG64(0x2201)).integers(0, 2**64, size=(1 << m, 3), dtype=np.uint64))
d_out = ops.empty(1 << m)
for _ in range(3):
    lib.additive_FFT_dev(d_in.data_ptr(), 1 << m, basis, shift, d_out.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    lib.additive_FFT_dev(d_in.data_ptr(), 1 << m, basis, shift, d_out.data_ptr())
torch.cuda.synchronize()
print("wall %.3f ms per transform" % ((time.perf_counter() - t0) * 100))
lib.profile_begin()
lib.additive_FFT_dev(d_in.data_ptr(), 1 << m, basis, shift, d_out.data_ptr())
rep = lib.profile_report()
tot = 0.0
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print("%-28s x%-3d %.3f ms" % (k, v[0], v[1]))
    tot += v[1]
print("sum %.3f ms in %d launches" % (tot, sum(v[0] for v in rep.values())))
