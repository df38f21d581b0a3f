"""Per-kernel profile of the FRI-only SNARK (BASELINE configs[2]: degree 2^20 on the 2^22-point domain over GF(2^192)) through iopx_fri_snark_prove on cuda:0."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, libiop_amd
from libiop_amd import domains, r1cs
lib = libiop_amd.lib(); lib.init(0); lib.set_stream(torch.cuda.current_stream().cuda_stream)
field = domains.GF192()
ops = domains.DeviceOps(lib, torch, torch.device("cuda:0"), field)
c = ops.upload(r1cs.seeded_elements(field, 0x2203, 1 << 20))
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr = lib.fri_snark_prove(0, c.data_ptr(), 1 << 20, 22, 2, 2, 1, 10)
    torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) * 1e3)
lib.profile_begin()
lib.fri_snark_prove(0, c.data_ptr(), 1 << 20, 22, 2, 2, 1, 10)
p = lib.profile_report()
print("total", sum(v[1] for v in p.values()))
for k, v in sorted(p.items(), key=lambda kv: -kv[1][1])[:14]: print("  %-30s %3d %7.3f" % (k, v[0], v[1]))
