"""Fractal indexer and prover on one MI355X at a given size (BASELINE config 5: 2^20 constraints over the 181-bit field): wall-clock
of the indexer, of the prover per round, and per kernel (library HIP events)."""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import libiop_amd
from libiop_amd import domains, fractal, r1cs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--field", default="edwards_Fr")
    ap.add_argument("--inputs", type=int, default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check the last transcript with the oracle verifier (needs oracle/liboracle.so)")
    ap.add_argument("--force-sharded", action="store_true", help="multi-GPU operator set with one rank (exercises the RCCL calls on a 1-GPU box)")
    ap.add_argument("--native", action="store_true", help="also time the native prover (iopx_fractal_index / _prove; their _dist forms over an RCCL communicator "
                    "when run under torch.distributed.run or with --force-sharded) and compare its transcript with the Python prover's")
    ap.add_argument("--native-only", action="store_true", help="with --native: skip the Python prover (profiling runs); the indexer still runs once for the roots")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N tools/fractal_bench.py ...` (RCCL); alone otherwise
    rank, local_rank, world = (int(os.environ.get(v, d)) for v, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world > 1 or a.force_sharded:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    lib = libiop_amd.lib()
    lib.init(local_rank)
    dev = torch.device("cuda", local_rank)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    field = domains.GF192() if a.field == "gf192" else domains.EdwardsFr()
    k = a.inputs if a.inputs is not None else (15 if field.additive else 0)            # instrument_fractal_snark.cpp:104-110
    if world > 1 or a.force_sharded:
        from libiop_amd import dist as idist
        ops = idist.sharded_ops(lib, torch, dev, field, idist.AuroraShard(dist, rank, world))   # residue classes for the prime field
    else:
        ops = domains.DeviceOps(lib, torch, dev, field)
    n = 1 << a.log_n
    t0 = time.time()
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, k, n - 1, 0x2205)
    torch.cuda.synchronize()
    print("instance generated in %.2f s" % (time.time() - t0), flush=True)
    params = fractal.FractalParameters(field, cs)
    res = {"log_n": a.log_n, "field": a.field, "num_inputs": k, "n_gpus": world, "codeword_domain_dim": params.codeword_domain_dim,
           "fri_query_repetitions": params.fri_query_repetitions, "localization_parameters": params.localization_parameters, "runs": []}
    for rep in range(1 if a.native_only else 2):
        torch.cuda.synchronize()
        t0 = time.time()
        index, (roots, _) = fractal.fractal_snark_indexer(ops, cs, params)
        torch.cuda.synchronize()
        res.setdefault("indexer_s", []).append(time.time() - t0)
    if rank == 0:
        print("indexer: %s s" % res["indexer_s"], flush=True)
    tr = None
    for rep in range(0 if a.native_only else a.reps):
        marks = []
        torch.cuda.synchronize()
        if a.profile and rep == a.reps - 1:
            lib.profile_begin()
        t0 = time.time()
        def hook(r):
            lib.synchronize()
            marks.append((r, time.time() - t0))
        tr = fractal.fractal_snark_prover(ops, index, cs, primary, auxiliary, params, round_hook=hook)
        torch.cuda.synchronize()
        total = time.time() - t0
        run = {"prover_s": total, "round_marks": marks, "argument_bytes": len(tr.serialize())}
        if a.profile and rep == a.reps - 1:
            prof = lib.profile_report()
            run["kernels"] = {kk: {"launches": v[0], "ms": v[1], "bytes": v[2]} for kk, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        res["runs"].append(run)
        if rank == 0:
            print(json.dumps(run)[:3000], flush=True)
    if a.native:
        comm = lib.comm_create_rccl_from_torch(dist, rank, world, dev) if (world > 1 or a.force_sharded) else None
        inst = lib.aurora_example_instance(0 if field.additive else 1, n, k, n - 1, 0x2205)
        torch.cuda.synchronize()
        t0 = time.time()
        nroots = lib.fractal_index_dist(inst, comm) if comm is not None else lib.fractal_index(inst)
        torch.cuda.synchronize()
        res["native_indexer_s"] = time.time() - t0
        times = []
        for rep in range(max(a.reps, 3)):
            torch.cuda.synchronize()
            t0 = time.time()
            nt = lib.fractal_prove_dist(inst, comm) if comm is not None else lib.fractal_prove(inst)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
        lib.comm_stats(reset=True)
        lib.profile_begin()
        (lib.fractal_prove_dist(inst, comm) if comm is not None else lib.fractal_prove(inst))
        nprof = lib.profile_report()
        if a.native_only:             # the profiled proof in the place tools/make_traffic_json.py reads
            res["runs"].append({"prover_s": min(times), "argument_bytes": len(nt), "prover": "native",
                                "kernels": {kk: {"launches": v[0], "ms": v[1], "bytes": v[2]} for kk, v in sorted(nprof.items(), key=lambda kv: -kv[1][1])}})
        res["native"] = {"prover_s": times, "prover_s_min": min(times), "index_roots_equal": [bytes(r) for r in roots] == nroots,
                         "transcript_equals_python": (nt == tr.serialize()) if tr is not None else None,
                         "collectives_per_proof": lib.comm_stats()[0], "collective_bytes_per_proof_this_rank": lib.comm_stats()[1],
                         "path": "iopx_fractal_prove_dist over an RCCL communicator of %d rank(s)" % world if comm is not None else "iopx_fractal_prove",
                         "kernels_ms_total": sum(v[1] for v in nprof.values()),
                         "kernels": {kk: {"launches": v[0], "ms": round(v[1], 4), "bytes": v[2]} for kk, v in sorted(nprof.items(), key=lambda kv: -kv[1][1])}}
        lib.aurora_instance_free(inst)
        if comm is not None:
            lib.comm_destroy(comm)
        if rank == 0:
            print("native:", json.dumps(res["native"]), flush=True)
    if a.verify and rank == 0:
        import oracle
        code = oracle.FIELD_GF192 if field.additive else oracle.FIELD_EDWARDS
        t0 = time.time()
        ok = oracle.fractal_verify(code, a.log_n, k, 0x2205, tr.serialize(), [bytes(r) for r in roots])
        res["oracle_verifier_accepts"] = bool(ok)
        print("oracle verifier: %s (%.1f s)" % (ok, time.time() - t0), flush=True)
    if a.out and rank == 0:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


    if world > 1 or a.force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
