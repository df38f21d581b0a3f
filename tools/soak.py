#!/usr/bin/env python3
"""Soak test of the native Aurora prover on cuda:0: 301 proofs of the 2^20 instance, every transcript equal to the first, free HBM printed every
100 proofs (a leak in the buffer pool or the pinned staging would show as a falling figure)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libiop_amd
lib = libiop_amd.Library()
lib.set_stream(torch.cuda.current_stream().cuda_stream)
inst = lib.aurora_example_instance(0, 1 << 20, 15, (1 << 20) - 1, 0x20)
first = None
t0 = time.time()
for i in range(301):
    t = lib.aurora_prove(inst, 128, 5, 2)
    if first is None:
        first = bytes(t)
    assert bytes(t) == first
    if i % 100 == 0:
        free, total = torch.cuda.mem_get_info()
        print(i, "free GB %.3f" % (free / 2**30), "elapsed %.1f s" % (time.time() - t0), flush=True)
lib.aurora_instance_free(inst)
print("ok", len(first))
