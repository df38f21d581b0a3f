#!/usr/bin/env python3
"""Every rank of the 2-, 4- and 8-rank Aurora prover played alone on this GPU (bench.rank_replay over iopx_comm_create_replay): the per-rank compute path of the
2^20 proof, collectives completed locally (not timed).  Writes one JSON: {world: [per-rank entries]} and the max over the ranks of each world — a lower bound
on the proof's time over that many GPUs.   python tools/rank_replay_all.py [--log-n 20] [--out FILE]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import libiop_amd
from libiop_amd import aurora, domains

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--out", default=None)
a = ap.parse_args()
lib = libiop_amd.lib()
lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << a.log_n
params = aurora.AuroraParameters(domains.GF192(), n, n - 1, 15)
inst = lib.aurora_example_instance(0, n, 15, n - 1, bench.SEED)
lib.aurora_instance_warm(inst)
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    lib.aurora_prove(inst)
torch.cuda.synchronize()
single = (time.perf_counter() - t0) / 5 * 1e3
out = {"log_n": a.log_n, "single_gpu_ms": round(single, 3), "what": "bench.rank_replay: compute path per rank, collectives completed locally (not timed)", "worlds": {}}
for w in (2, 4, 8):
    ranks = [bench.rank_replay(lib, torch, inst, params, w, r, steps=4, warmup=1) for r in range(w)]
    worst = max(x["ms_per_proof_pow_adjusted"] for x in ranks)
    out["worlds"][str(w)] = {"max_ms_pow_adjusted": worst, "single_over_max": round(single / worst, 2),
                             "ranks": [{k: x[k] for k in ("rank", "ms_per_proof", "ms_per_proof_pow_adjusted", "kernels_ms_total", "collectives_per_proof", "collective_payload_bytes_this_rank")} for x in ranks]}
    print(w, [x["ms_per_proof_pow_adjusted"] for x in ranks], flush=True)
lib.aurora_instance_free(inst)
text = json.dumps(out, indent=1)
if a.out:
    open(a.out, "w").write(text)
print(text)
