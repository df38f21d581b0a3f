#!/usr/bin/env python3
"""One BLAKE2b Merkle tree over `--oracles` columns of 2^log_n GF(2^192) elements, cosets of 2, device-resident: per-kernel
HIP-event times (run under rocprofv3 --kernel-trace + tools/kernel_launches.py for per-launch durations)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import libiop_amd

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
r = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = libiop_amd.lib()
lib.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
cols = [torch.randint(-2**63, 2**63 - 1, (1 << log_n, 3), dtype=torch.int64, device=dev, generator=g) for _ in range(r)]
L = 1 << (log_n - 1)
nodes = torch.empty((2 * L - 1, 32), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
run = lambda: lib.merkle_tree_dev([c.data_ptr() for c in cols], 24, 1 << log_n, 2, nodes.data_ptr())
run()
lib.synchronize()
lib.profile_begin()
for _ in range(3):
    run()
rep = lib.profile_report()
print(json.dumps({"log_n": log_n, "oracles": r, **{k: (v[0] // 3, round(v[1] / 3, 3)) for k, v in rep.items()}}))
