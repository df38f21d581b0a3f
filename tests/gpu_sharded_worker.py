"""Worker of tests/test_gpu_sharded.py: one rank of a `python -m torch.distributed.run` job on the MI355X box (backend nccl = RCCL).
Runs every case of --cases (a JSON list of {protocol, field, impl, log_n, inputs, seed}) in ONE process group — a child per case spent most of
its time importing torch and creating the group — with the multi-GPU operator sets, and writes what rank 0 produced to --out as a JSON list
(hex transcripts), one entry per case, in order.

  impl python   libiop_amd/dist.py: ShardedDeviceOps (GF(2^192), contiguous cosets) / ResidueShardedDeviceOps (181-bit field, residue classes)
  impl native   iopx_aurora_prove_dist / iopx_fractal_{index,prove}_dist: the C++ prover inside the library, RCCL communicator through the C ABI
  protocol fft  iopx_add_[i]fft_gf192_dist_dev: one transform as long as its domain, block-distributed over the ranks
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_fft(a, lib, torch, dist, dev, rank, world, comm):
    import numpy as np
    import libiop_amd
    m = a.log_n
    basis = libiop_amd.standard_basis(m)
    shift = np.array([1 << m, 0, 0], dtype=np.uint64)
    coeffs = np.random.Generator(np.random.PCG64(a.seed)).integers(0, 2**64, size=(1 << m, 3), dtype=np.uint64)
    d_all = torch.from_numpy(coeffs.view(np.int64)).to(dev)
    d_full = torch.empty_like(d_all)
    lib.additive_FFT_dev(d_all.data_ptr(), 1 << m, basis, shift, d_full.data_ptr())
    per = (1 << m) // world
    mine, back = torch.empty((per, 3), dtype=torch.int64, device=dev), torch.empty((per, 3), dtype=torch.int64, device=dev)
    block = d_all[rank * per:(rank + 1) * per].contiguous()
    lib.additive_FFT_dist_dev(comm, block.data_ptr(), basis, shift, mine.data_ptr())
    lib.additive_FFT_dist_dev(comm, mine.data_ptr(), basis, shift, back.data_ptr(), inverse=True)
    torch.cuda.synchronize()
    ok = bool(torch.equal(mine, d_full[rank * per:(rank + 1) * per]) and torch.equal(back, block))
    flags = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([int(ok)], dtype=torch.int64, device=dev))
    return {"world": world, "ranks_agree": True, "fft_ok": [int(x.item()) for x in flags]}


def run_prover(a, lib, torch, dist, dev, rank, world, comm):
    from libiop_amd import aurora, domains, fractal, r1cs
    from libiop_amd import dist as idist
    field = domains.GF192() if a.field == "gf192" else domains.EdwardsFr()
    n = 1 << a.log_n
    res = {"world": world, "impl": a.impl}
    if a.impl == "python":
        ops = idist.sharded_ops(lib, torch, dev, field, idist.AuroraShard(dist, rank, world))
        res["ops"] = type(ops).__name__
        cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, a.inputs, n - 1, a.seed)
        d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
        if a.protocol == "aurora":
            params = aurora.AuroraParameters(field, n, n - 1, a.inputs)
            t = idist.sharded_aurora_snark_prover(ops, cs, primary, params, d_z).serialize()
            roots = []
        else:
            params = fractal.FractalParameters(field, cs)
            index, (roots, _) = idist.sharded_fractal_snark_indexer(ops, cs, params)
            t = idist.sharded_fractal_snark_prover(ops, index, cs, primary, params, d_z).serialize()
            roots = [bytes(r).hex() for r in roots]
    else:
        inst = lib.aurora_example_instance(0 if a.field == "gf192" else 1, n, a.inputs, n - 1, a.seed)
        try:
            if a.protocol == "aurora":
                t = lib.aurora_prove_dist(inst, comm)
                roots = []
            else:
                roots = [r.hex() for r in lib.fractal_index_dist(inst, comm)]
                t = lib.fractal_prove_dist(inst, comm)
        finally:
            lib.aurora_instance_free(inst)
    torch.cuda.synchronize()
    # every rank must hold the same transcript: compare digests across the ranks
    h = torch.tensor(list(hashlib.blake2b(t, digest_size=32).digest()), dtype=torch.uint8, device=dev)
    hs = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    res["ranks_agree"] = all(bool(torch.equal(x, h)) for x in hs)
    if rank == 0:
        # the single-GPU native prover on the same seeded instance, for sizes at which the oracle prover takes minutes
        inst = lib.aurora_example_instance(0 if a.field == "gf192" else 1, n, a.inputs, n - 1, a.seed)
        try:
            if a.protocol == "aurora":
                single = lib.aurora_prove(inst)
            else:
                lib.fractal_index(inst)
                single = lib.fractal_prove(inst)
        finally:
            lib.aurora_instance_free(inst)
        res.update({"transcript": t.hex(), "index_roots": roots, "equals_single_gpu_native_prover": single == t})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", required=True, help="JSON list of {protocol, field, impl, log_n, inputs, seed}")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    cases = [argparse.Namespace(**c) for c in json.loads(args.cases)]

    import torch
    import torch.distributed as dist
    import libiop_amd

    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = libiop_amd.lib()
    lib.init(local_rank)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    comm = lib.comm_create_rccl_from_torch(dist, rank, world, dev)          # one RCCL communicator through the C ABI for every native case
    results = []
    try:
        for a in cases:
            try:
                results.append((run_fft if a.protocol == "fft" else run_prover)(a, lib, torch, dist, dev, rank, world, comm))
            except Exception as e:                                          # reported per case; the cases after it still run
                results.append({"world": world, "error": "%s: %s" % (type(e).__name__, e)})
        if rank == 0:
            with open(args.out, "w") as f:
                json.dump(results, f)
    finally:
        lib.comm_destroy(comm)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
