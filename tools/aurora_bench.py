"""Aurora prover on one MI355X at a given size: wall-clock per prover round and per kernel (library HIP events)."""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import libiop_amd
from libiop_amd import aurora, domains, r1cs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--field", default="gf192")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = libiop_amd.lib()
    lib.init(0)
    dev = torch.device("cuda:0")
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    field = domains.GF192() if a.field == "gf192" else domains.EdwardsFr()
    ops = domains.DeviceOps(lib, torch, dev, field)
    n = 1 << a.log_n
    t0 = time.time()
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, 15, n - 1, 0x2204)
    torch.cuda.synchronize()
    print("instance generated in %.2f s" % (time.time() - t0), flush=True)
    params = aurora.AuroraParameters(field, n, n - 1, 15)
    res = {"log_n": a.log_n, "field": a.field, "runs": []}
    for rep in range(a.reps):
        marks = []
        torch.cuda.synchronize()
        if a.profile and rep == a.reps - 1:
            lib.profile_begin()
        t0 = time.time()
        def hook(r):
            lib.synchronize()
            marks.append((r, time.time() - t0))
        tr = aurora.aurora_snark_prover(ops, cs, primary, auxiliary, params, round_hook=hook)
        torch.cuda.synchronize()
        total = time.time() - t0
        run = {"prover_s": total, "round_marks": marks, "argument_bytes": len(tr.serialize())}
        if a.profile and rep == a.reps - 1:
            prof = lib.profile_report()
            run["kernels"] = {k: {"launches": v[0], "ms": v[1]} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        res["runs"].append(run)
        print(json.dumps(run)[:2000], flush=True)
        del tr
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
