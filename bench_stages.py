#!/usr/bin/env python3
"""Stage benchmarks of the prover hot path beyond bench.py's headline transform.

  --config cfg3   FRI prover commit phase (protocols/ldt/fri) on a degree-2^20 RS codeword over GF(2^192), 2^22-point
                  standard-basis domain, localization array [1,2x9], incl. the per-round BLAKE2b Merkle trees, 1 GPU.
  --config cfg4   Aurora 2^20 over GF(2^192) as a STAGE REPLAY (SURVEY.md §3.1 / §8d inventory; the protocol logic that
                  produces the real inputs is a 'next' row): 8 codeword LDEs 2^20 -> 2^25, IFFTs 6 x 2^20 + 1 x 2^21,
                  Merkle round 0 (4 oracles, cosets of 2) and round 1 (1 oracle), FRI commit from 2^25.  With
                  --gpus N (torchrun) every oracle is sharded by contiguous blocks (libiop_amd/dist.py, no data-path
                  collective; N sub-roots all-gathered per tree).
Prints one JSON line with per-stage milliseconds (wall, synchronised) and the per-kernel HIP-event times.
--cpu runs the same stages through the CPU oracle (single thread) on a bounded sample for the 'vs CPU' column."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg4"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--log-degree", type=int, default=20)
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on a bounded sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import libiop_amd
    from libiop_amd import dist as idist, fri, host

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = libiop_amd.lib()
    lib.init(local_rank)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)

    d = args.log_degree
    rs = 2 if args.config == "cfg3" else 5
    m = d + rs
    basis = libiop_amd.standard_basis(m)
    shift = np.zeros(3, dtype=np.uint64) if args.config == "cfg3" else np.array([1 << m, 0, 0], dtype=np.uint64)
    loc = host.localization_parameter_to_array(2, m, rs)
    stages = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        stages[name] = stages.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return out

    def rand_dev(n, seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device=dev, generator=g)

    lib.profile_begin()
    t_all = time.perf_counter()
    if args.config == "cfg3":
        coeffs = rand_dev(1 << d, 0x2203)
        cw = torch.empty((1 << m, 3), dtype=torch.int64, device=dev)
        # warm the per-domain plan (tables) outside the timed stages, as a prover reuses its domains
        lib.additive_FFT_dev(coeffs.data_ptr(), 1 << d, basis, shift, cw.data_ptr())
        timed("lde_fft_2^%d->2^%d" % (d, m), lambda: lib.additive_FFT_dev(coeffs.data_ptr(), 1 << d, basis, shift, cw.data_ptr()))
        doms = host.fri_additive_domains(basis, shift, loc)      # protocol setup (fri_ldt.tcc:279-340), not prover time
        fri.fri_commit(lib, torch, cw, basis, shift, loc, final_degree_bound=1 << (d - sum(loc)), domains=doms)   # warm-up
        res = timed("fri_commit(merkle+fold x%d, final ifft)" % len(loc),
                    lambda: fri.fri_commit(lib, torch, cw, basis, shift, loc, final_degree_bound=1 << (d - sum(loc)), domains=doms))
        extra = {"roots": [r.hex()[:16] for r in res.roots], "final_poly_len": int(res.final_polynomial.shape[0])}
    else:
        n_loc = (1 << m) // world
        coeffs = [rand_dev(1 << d, 0x2204 + k) for k in range(8)]
        lib.additive_LDE_dev(coeffs[0].data_ptr(), 1 << d, basis, shift, 0, 1, torch.empty((1 << d, 3), dtype=torch.int64, device=dev).data_ptr())
        cws = []
        for k in range(8):
            cws.append(timed("lde_fft_x8", lambda k=k: idist.sharded_lde(lib, torch, coeffs[k], (1 << d) - (16 if k == 0 else 0), basis, shift, rank, world)))
        if rank == 0:
            b20, b21 = libiop_amd.standard_basis(d), libiop_amd.standard_basis(d + 1)
            tmp = torch.empty((1 << (d + 1), 3), dtype=torch.int64, device=dev)
            lib.additive_IFFT_dev(cws[0].data_ptr(), b20, np.zeros(3, dtype=np.uint64), tmp.data_ptr())
            for k in range(6):
                timed("ifft_6x2^20+1x2^21", lambda k=k: lib.additive_IFFT_dev(cws[k].data_ptr(), b20, np.zeros(3, dtype=np.uint64), tmp.data_ptr()))
            if n_loc >= (1 << (d + 1)):
                lib.additive_IFFT_dev(cws[6].data_ptr(), b21, shift, tmp.data_ptr())
                timed("ifft_6x2^20+1x2^21", lambda: lib.additive_IFFT_dev(cws[6].data_ptr(), b21, shift, tmp.data_ptr()))
        r0 = timed("merkle_round0(4 oracles,c=2)", lambda: idist.sharded_merkle_root(lib, torch, dist, cws[:4], n_loc, 2, rank, world))[0]
        r1 = timed("merkle_round1(1 oracle,c=2)", lambda: idist.sharded_merkle_root(lib, torch, dist, cws[7:8], n_loc, 2, rank, world))[0]
        # FRI commit on the sharded codeword: per round Merkle (sub-roots gathered) + local fold; when a round's
        # codeword has fewer cosets than ranks the remainder is finished on rank 0 after an all-gather
        hc = host.Blake2bHashchain()
        doms = host.fri_additive_domains(basis, shift, loc)
        f, roots = cws[4], []

        def fri_rounds():
            nonlocal f
            cur_world, gathered = world, False
            for i, eta in enumerate(loc):
                b_i, s_i = doms[i]
                cs = 1 << eta
                if not gathered and world > 1 and f.shape[0] // cs < 2:
                    parts = [torch.empty_like(f) for _ in range(world)]
                    dist.all_gather(parts, f)
                    f = torch.cat(parts, 0)
                    gathered, cur_world = True, 1
                if gathered:
                    root, _ = idist.sharded_merkle_root(lib, torch, dist, [f], f.shape[0], cs, 0, 1)
                    x = None
                else:
                    root, _ = idist.sharded_merkle_root(lib, torch, dist, [f], f.shape[0], cs, rank, world)
                roots.append(root)
                hc.absorb(root); hc.absorb(None)
                x = hc.squeeze_gf192(1)[0]
                f = idist.sharded_fri_fold(lib, torch, f, b_i, s_i, cs, x, 0 if gathered else rank, cur_world)
        timed("fri_commit(merkle+fold x%d)" % len(loc), fri_rounds)
        extra = {"root0": r0.hex()[:16], "root1": r1.hex()[:16], "fri_roots": [r.hex()[:16] for r in roots]}
    total_ms = (time.perf_counter() - t_all) * 1e3
    prof = lib.profile_report()

    out = {"config": args.config, "n_gpus": world, "log_degree": d, "codeword_dim": m, "localization": loc,
           "stages_ms": stages, "stages_total_ms": sum(stages.values()), "wall_ms_incl_setup": total_ms,
           "kernels_ms": {k: round(v[1], 3) for k, v in prof.items()}, "kernel_launches": {k: v[0] for k, v in prof.items()},
           "replay": args.config == "cfg4", **extra}

    if args.cpu and rank == 0:
        import oracle
        # bounded CPU sample: the same stages at log-degree 16 (1/16 of the work per log step), single thread
        dd = min(d, 16)
        mm = dd + rs
        bb = oracle.standard_basis(mm, 3)
        ss = np.zeros(3, dtype=np.uint64)
        rng = np.random.Generator(np.random.PCG64(1))
        cc = rng.integers(0, 2**64, size=(1 << dd, 3), dtype=np.uint64)
        t0 = time.perf_counter(); cwc = oracle.additive_fft(cc, bb, ss); t_fft = time.perf_counter() - t0
        t0 = time.perf_counter(); oracle.merkle_build([cwc], 2, True); t_mt = time.perf_counter() - t0
        x = rng.integers(0, 2**64, size=3, dtype=np.uint64)
        t0 = time.perf_counter(); oracle.fri_fold_additive(cwc, bb, ss, 2, x); t_fold = time.perf_counter() - t0
        out["cpu_sample"] = {"log_degree": dd, "codeword_dim": mm, "cores": 1,
                             "fft_s": t_fft, "merkle_round0_s": t_mt, "fold_round0_s": t_fold,
                             "note": "reference-shaped CPU port (oracle, PCLMUL, 1 thread) at a 2^%d-point codeword; FFT cost scales "
                                     "as n*log^2 n (XOR sweeps) + 1.5 n log n products, Merkle and fold linearly" % mm}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
