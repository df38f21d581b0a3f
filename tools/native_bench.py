#!/usr/bin/env python3
"""Times the native Aurora prover (iopx_aurora_prove) on cuda:0 for one field and size, with the library's per-kernel profile of one proof.
Usage: native_bench.py [--field gf192|edwards_Fr] [--log-n 20] [--reps 5] [--out FILE]   (IOPX_HEAD_EVAL=0 in the environment: the reference's schedule)"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libiop_amd

ap = argparse.ArgumentParser()
ap.add_argument("--field", default="gf192")
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--out", default=None)
a = ap.parse_args()
lib = libiop_amd.lib()
lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << a.log_n
inst = lib.aurora_example_instance(0 if a.field == "gf192" else 1, n, 15, n - 1, 0x2204)
times = []
for rep in range(a.reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = lib.aurora_prove(inst)
    torch.cuda.synchronize()
    if rep >= 2:
        times.append(time.perf_counter() - t0)
lib.profile_begin()
lib.aurora_prove(inst)
prof = lib.profile_report()
res = {"field": a.field, "log_n": a.log_n, "schedule": "reference (IOPX_HEAD_EVAL=0)" if os.environ.get("IOPX_HEAD_EVAL", "1")[:1] == "0" else "head",
       "prover_ms": [round(x * 1e3, 3) for x in times], "prover_ms_min": round(min(times) * 1e3, 3), "argument_bytes": len(t),
       "transcript_blake2b": hashlib.blake2b(t, digest_size=32).hexdigest(), "kernels_ms_total": round(sum(v[1] for v in prof.values()), 3),
       "kernels": {k: {"launches": v[0], "ms": round(v[1], 4)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
lib.aurora_instance_free(inst)
print(json.dumps(res))
if a.out:
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
