#!/usr/bin/env python3
"""Stage benchmarks of the prover hot path beyond bench.py's headline transform.

  --config cfg5   Fractal 2^20 over the 181-bit prime field as a STAGE REPLAY: indexer (12 codeword FFTs 2^20 -> 2^25, one tree over 12
                  oracles), prover (8 codeword FFTs from 2^22 coefficients, known-degree IFFT), FRI commit; --gpus N shards by
                  residue class.
  --config cfg3   FRI prover commit phase (protocols/ldt/fri) on a degree-2^20 RS codeword over GF(2^192), 2^22-point
                  standard-basis domain, localization array [1,2x9], incl. the per-round BLAKE2b Merkle trees, 1 GPU.
  --config cfg4   Aurora 2^20 over GF(2^192) as a STAGE REPLAY (SURVEY.md §3.1 / §8d inventory; the protocol logic that
                  produces the real inputs is a 'next' row): 8 codeword LDEs 2^20 -> 2^25, IFFTs 6 x 2^20 + 1 x 2^21,
                  Merkle round 0 (4 oracles, cosets of 2) and round 1 (1 oracle), the virtual oracles (fz, row check, lincheck,
                  sumcheck g), the LDT-reducer combination of 7 oracles, FRI commit from 2^25, the proof of work and (one
                  GPU) the transcript of 32 queries.  With
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


def prewarm(torch, dev, count, nbytes):
    """Codeword-sized buffers are allocated inside the timed stages; take them from the driver once up front (a prover
    reuses its buffers; fresh VRAM after another process has to be scrubbed by the driver, which is not prover time)."""
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(count)]
    for b in bufs:
        b.zero_()
    torch.cuda.synchronize()
    del bufs


def cfg5(args, torch, dist, lib, dev, rank, world, la, fri, host):
    """Fractal 2^20 over the 181-bit prime field, multiplicative cosets, as a STAGE REPLAY:
    indexer = 12 codeword FFTs (2^20 coefficients -> 2^25-point coset, shift = multiplicative_generator) + one Merkle
    tree over the 12 oracles (cosets of 2, 576-byte leaves); prover = 8 codeword FFTs (2^22 coefficients), one strided
    IFFT of known degree, FRI commit [1,2x10] from 2^25.  With --gpus N (torchrun) every codeword is sharded by residue
    class mod N (libiop_amd/dist.py): LDE and folds are local, a tree needs one all-to-all of leaf digests and an all-gather of
    N sub-roots; when a FRI round has too few cosets to stay sharded the rest is gathered and finished on every rank."""
    from libiop_amd import dist as idist
    d, m = args.log_degree, args.log_degree + 5
    P = la.EDWARDS_FR_MODULUS
    shift_int = la.EDWARDS_FR_GENERATOR
    gen_int = pow(la.EDWARDS_FR_GENERATOR, (P - 1) >> m, P)
    n_loc = (1 << m) // world
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

    def rand_fp(n, seed):          # uniformly random canonical values < 2^180 < p, then to Montgomery form on the host
        rng = np.random.Generator(np.random.PCG64(seed))
        raw = rng.integers(0, 2**60, size=(n, 3), dtype=np.uint64)
        return torch.from_numpy(raw.view(np.int64)).to(dev)     # any 180-bit words are valid Montgomery representatives

    def fft(c, ncoef):
        return idist.sharded_mul_lde(lib, torch, la, c, ncoef, m, gen_int, shift_int, rank, world)

    prewarm(torch, dev, 14, n_loc * 24)
    lib.profile_begin()
    c20 = [rand_fp(1 << d, 50 + k) for k in range(12)]
    warm = fft(c20[0], 1 << d)            # builds the twiddle cache of the rank's coset (per domain, like subgroup.tcc:117-144)
    del warm
    idx = [timed("indexer_fft_x12(2^%d->2^%d)" % (d, m), lambda k=k: fft(c20[k], 1 << d)) for k in range(12)]
    root, _ = timed("indexer_merkle(12 oracles,c=2)", lambda: idist.sharded_mul_merkle_root(lib, torch, dist, la, idx, n_loc, 2, rank, world))
    del idx[4:]
    c22 = rand_fp(1 << (d + 2), 99)
    cws = [timed("prover_fft_x8(2^%d->2^%d)" % (d + 2, m), lambda: fft(c22, (1 << (d + 2)) - 1)) for _ in range(2)]
    for _ in range(6):
        timed("prover_fft_x8(2^%d->2^%d)" % (d + 2, m), lambda: fft(c22, (1 << (d + 2)) - 1))
    # IFFT_of_known_degree reads every (n / 2^ceil(log degree))-th evaluation (fft.tcc:450-454): a multiple of N apart for N <= 8,
    # i.e. all inside residue class 0
    lg_loc, g_loc, s_loc = idist.local_coset(m, gen_int, shift_int, 0, world, P)
    if rank == 0:
        co = torch.empty((1 << (d + 2), 3), dtype=torch.int64, device=dev)
        kd = lambda: lib._check(lib.c.iopx_mul_ifft_known_degree_fp3_dev(
            cws[0].data_ptr(), (1 << (d + 2)) - 1, lg_loc, la._as_u64(la.edwards_to_montgomery([g_loc])[0]).ctypes.data_as(la._u64p),
            la._as_u64(la.edwards_to_montgomery([s_loc])[0]).ctypes.data_as(la._u64p), co.data_ptr()))
        kd()
    timed("ifft_known_degree(2^%d of 2^%d)" % (d + 2, m), lambda: kd() if rank == 0 else None)
    loc = host.localization_parameter_to_array(2, m, 3)

    def fri_commit():
        return idist.sharded_mul_fri_commit(lib, torch, dist, la, fri, cws[1], m, gen_int, shift_int, loc, 4, rank, world)[0]

    fri_commit()        # warm-up: twiddle caches of the round domains
    roots = timed("fri_commit(merkle+fold x%d, final ifft)" % len(loc), fri_commit)
    prof = lib.profile_report()
    if rank == 0:
        print(json.dumps({"config": "cfg5", "n_gpus": world, "log_degree": d, "codeword_dim": m, "localization": loc, "replay": True,
                          "stages_ms": stages, "stages_total_ms": sum(stages.values()),
                          "kernels_ms": {k: round(v[1], 3) for k, v in prof.items()}, "index_root": root.hex()[:16],
                          "fri_roots": [r.hex()[:16] for r in roots]}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg4", "cfg5"])
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

    if args.config == "cfg5":
        return cfg5(args, torch, dist, lib, dev, rank, world, libiop_amd, fri, host)
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
        prewarm(torch, dev, 16, n_loc * 24)
        coeffs = [rand_dev(1 << d, 0x2204 + k) for k in range(8)]
        lib.additive_LDE_dev(coeffs[0].data_ptr(), 1 << d, basis, shift, 0, 1, torch.empty((1 << d, 3), dtype=torch.int64, device=dev).data_ptr())
        lib.synchronize()
        # the eight codewords of the round share one domain: one batched call (phase 1 of the transforms runs once)
        cws = timed("lde_fft_x8", lambda: idist.sharded_lde_batch(lib, torch, coeffs, 1 << d, basis, shift, rank, world))
        # interpolation back to coefficients touches only the first 2^20 / 2^21 evaluations (rank 0's block); the stage is timed
        # on every rank (timed() is collective)
        b20, b21 = libiop_amd.standard_basis(d), libiop_amd.standard_basis(d + 1)
        z3 = np.zeros(3, dtype=np.uint64)
        if rank == 0:
            tmp = torch.empty((6 << d, 3), dtype=torch.int64, device=dev)
            tmp21 = torch.empty((1 << (d + 1), 3), dtype=torch.int64, device=dev)

        def ifft6():            # six interpolations over the same 2^20-point domain, batched (inputs gathered back to back)
            if rank != 0:
                return
            src = torch.cat([cws[k][: 1 << d] for k in range(6)], 0)
            idist._torch_sync(torch, src)
            lib.additive_IFFT_batch_dev(src.data_ptr(), 6, b20, z3, tmp.data_ptr())

        def ifft21():
            if rank == 0 and n_loc >= (1 << (d + 1)):
                lib.additive_IFFT_dev(cws[6].data_ptr(), b21, shift, tmp21.data_ptr())
        ifft6()
        ifft21()
        timed("ifft_6x2^20+1x2^21", ifft6)
        timed("ifft_6x2^20+1x2^21", ifft21)
        r0 = timed("merkle_round0(4 oracles,c=2)", lambda: idist.sharded_merkle_root(lib, torch, dist, cws[:4], n_loc, 2, rank, world))[0]
        r1 = timed("merkle_round1(1 oracle,c=2)", lambda: idist.sharded_merkle_root(lib, torch, dist, cws[7:8], n_loc, 2, rank, world))[0]
        # virtual oracles evaluated over the whole codeword domain before the LDT reducer reads them (r1cs_rs_iop.tcc:181-222,
        # rowcheck.tcc:16-88, basic_lincheck_aux.tcc:102-144, sumcheck.tcc:58-119); pointwise, so a rank works on its block over
        # its local sub-domain.  Inputs are the replay's codewords (right sizes, synthetic values).
        b_loc, s_loc = idist.local_subdomain(basis, shift, rank, world)
        zero3, mu = np.zeros(3, dtype=np.uint64), np.array([5, 6, 7], dtype=np.uint64)
        vo_out = [torch.empty_like(cws[0]) for _ in range(4)]
        r3 = np.random.default_rng(8).integers(0, 2**63, size=(3, 3), dtype=np.uint64)

        def virtual_oracles():
            lib.fz_dev(cws[0].data_ptr(), cws[5].data_ptr(), b_loc, s_loc, basis[:4], zero3, vo_out[0].data_ptr())
            lib.rowcheck_dev(cws[1].data_ptr(), cws[2].data_ptr(), cws[3].data_ptr(), b_loc, s_loc, d, zero3, vo_out[1].data_ptr())
            lib.lincheck_dev(vo_out[0].data_ptr(), [cws[1].data_ptr(), cws[2].data_ptr(), cws[3].data_ptr()], r3, cws[5].data_ptr(),
                             cws[6].data_ptr(), n_loc, vo_out[2].data_ptr())
            lib.sumcheck_g_dev(vo_out[2].data_ptr(), cws[7].data_ptr(), b_loc, s_loc, basis[:d], zero3, mu, vo_out[3].data_ptr())
        virtual_oracles()
        timed("virtual_oracles(fz, rowcheck, lincheck, sumcheck g)", virtual_oracles)
        del vo_out
        # LDT reducer: the random combination of the round's oracles that FRI is run on (ldt_reducer_aux.tcc:39-131); pointwise, so
        # a rank combines its own block over its local sub-domain.  Aurora-like degree spread (one maximal oracle).
        degrees = [(1 << (d + 1)) - 1, 1 << d, 1 << d, 1 << d, (1 << d) + 2 * 41 - 1, (1 << (d + 1)) - 2, (1 << d) - 1]
        rcoef = np.random.default_rng(7).integers(0, 2**63, size=(2 * len(degrees), 3), dtype=np.uint64)
        comb = torch.empty_like(cws[0])
        ldt = lambda: lib.ldt_combine_dev([c.data_ptr() for c in cws[:7]], degrees, rcoef, b_loc, s_loc, comb.data_ptr())
        ldt()
        timed("ldt_combine(7 oracles)", ldt)
        # FRI commit on the sharded codeword: per round Merkle (sub-roots gathered) + local fold; when a round's
        # codeword has fewer cosets than ranks the remainder is finished on rank 0 after an all-gather
        hc = host.Blake2bHashchain()
        doms = host.fri_additive_domains(basis, shift, loc)
        f, roots, kept = comb, [], []

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
                    root, nodes = idist.sharded_merkle_root(lib, torch, dist, [f], f.shape[0], cs, 0, 1)
                else:
                    root, nodes = idist.sharded_merkle_root(lib, torch, dist, [f], f.shape[0], cs, rank, world)
                roots.append(root)
                kept.append((f, nodes, cs))
                hc.absorb(root); hc.absorb(None)
                x = hc.squeeze_gf192(1)[0]
                f = idist.sharded_fri_fold(lib, torch, f, b_i, s_i, cs, x, 0 if gathered else rank, cur_world)
        timed("fri_commit(merkle+fold x%d)" % len(loc), fri_rounds)
        # proof of work at the config's difficulty (dim_h + 3 bits, common_bcs_parameters.tcc:23-25), then — single GPU — the
        # transcript: query positions, pruned membership proofs and the queried cosets of every FRI round
        hc.absorb(None)
        challenge = hc.squeeze_root_type()
        answer = timed("pow(%d bits)" % (d + 3), lambda: lib.solve_pow(challenge, d + 3) if rank == 0 else None)
        extra = {"root0": r0.hex()[:16], "root1": r1.hex()[:16], "fri_roots": [r.hex()[:16] for r in roots]}
        if world == 1:
            hc.absorb(answer)
            num_queries = 32
            positions = fri.fri_query_positions(hc, num_queries, 1 << m)

            def transcript():
                sizes, sb = [], 0
                for (f_i, nodes, cs), eta in zip(kept, loc):
                    sb += eta
                    leaves = sorted(set(p >> sb for p in positions))
                    vals = lib.query_responses_dev([f_i.data_ptr()], 24, f_i.shape[0], [l * cs + k for l in leaves for k in range(cs)])
                    aux = lib.get_set_membership_proof_dev(nodes.data_ptr(), f_i.shape[0] // cs, leaves)
                    sizes.append(vals.nbytes + aux.nbytes)
                return sizes
            transcript()        # warm-up: first use loads the gather kernels and sizes the library's temporaries
            sizes = timed("transcript(%d queries: responses + membership proofs)" % num_queries, transcript)
            extra["fri_transcript_bytes"] = int(sum(sizes))
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
