#!/usr/bin/env python3
"""Soak test of the native Fractal prover on cuda:0 (181-bit field, 2^20 constraints): index once, 101 proofs, every transcript equal to the first and to the
oracle's digest (tests/golden/oracle_fractal_transcript_digests_large.json), free HBM printed every 50 proofs.  Round 5: the trees of every round are
built on the side stream, beside the next round's kernels — a lost ordering between the two streams would show as a transcript that differs once."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libiop_amd

lib = libiop_amd.Library()
lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << 20
want = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_fractal_transcript_digests_large.json")))["digests"]["20"]
inst = lib.aurora_example_instance(1, n, 0, n - 1, 0x2205)
roots = lib.fractal_index(inst)
assert [r.hex() for r in roots] == want["index_roots"]
first = None
t0 = time.time()
for i in range(101):
    t = lib.fractal_prove(inst)
    if first is None:
        first = bytes(t)
        assert hashlib.blake2b(first, digest_size=32).hexdigest() == want["transcript_blake2b"]
    assert bytes(t) == first
    if i % 50 == 0:
        free, total = torch.cuda.mem_get_info()
        print(i, "free GB %.3f" % (free / 2**30), "elapsed %.1f s" % (time.time() - t0), flush=True)
lib.aurora_instance_free(inst)
print("ok", len(first))
