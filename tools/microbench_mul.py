#!/usr/bin/env python3
"""Field-multiplication micro-benchmark on the GPU: general (per-lane) vs wave-uniform multiplier."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import libiop_amd

lib = libiop_amd.lib(); lib.init(0)
n = 1 << 24
rng = np.random.Generator(np.random.PCG64(1))
a = rng.integers(0, 2**64, size=(n, 3), dtype=np.uint64)
da, db, do = lib.malloc(a.nbytes), lib.malloc(a.nbytes), lib.malloc(a.nbytes)
lib.h2d(da, a); lib.h2d(db, a[::-1].copy())
for name, fn in (("general", lambda: lib.gf192_mul_dev(da, db, do, n)),
                 ("uniform", lambda: lib._check(lib.c.iopx_gf192_mul_uniform_dev(da, db, do, n)))):
    fn(); lib.synchronize()
    lib.profile_begin()
    for _ in range(5):
        fn()
    rep = lib.profile_report()
    for k, (cnt, ms, _bytes) in rep.items():
        print("%s: %s %.3f ms/launch  %.3e mult/s  (%.1f GB/s)" % (name, k, ms / cnt, n / (ms / cnt / 1e3), 3 * a.nbytes / (ms / cnt / 1e3) / 1e9))
