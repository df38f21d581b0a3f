#!/usr/bin/env python3
"""Per-kernel timing of one 2^20 additive IFFT and one 2^20 FFT over GF(2^192) (standard basis), device-resident."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import libiop_amd

lib = libiop_amd.lib()
lib.init(0)
dev = torch.device("cuda", 0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
basis, shift = libiop_amd.standard_basis(m), np.zeros(3, dtype=np.uint64)
g = torch.Generator(device=dev)
g.manual_seed(1)
c = torch.randint(-2**63, 2**63 - 1, (1 << m, 3), dtype=torch.int64, device=dev, generator=g)
o = torch.empty_like(c)
torch.cuda.synchronize()
out = {}
for name, fn in (("fft", lambda: lib.additive_FFT_dev(c.data_ptr(), 1 << m, basis, shift, o.data_ptr())),
                 ("ifft", lambda: lib.additive_IFFT_dev(c.data_ptr(), basis, shift, o.data_ptr()))):
    fn()
    lib.synchronize()
    lib.profile_begin()
    for _ in range(5):
        fn()
    rep = lib.profile_report()
    out[name] = {"total_ms": round(sum(v[1] for v in rep.values()) / 5, 3), **{k: (v[0] // 5, round(v[1] / 5, 3)) for k, v in rep.items()}}
print(json.dumps(out))
