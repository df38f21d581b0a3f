"""world_size-2 checks of the multi-GPU sharding (libiop_amd/dist.py) with the gloo backend on CPU: each rank
runs the kernels through the CPU emulation (tests/emu) and the union of the shards must equal the oracle's
single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from emu_lib import emu
        from helpers import rand_elems
        from libiop_amd import dist as idist
        lib = emu()
        m, ncoef, cs = 9, 50, 2            # d = 6 -> 8 cosets, 4 per rank
        basis = oracle.standard_basis(m, W)
        shift = np.array([1 << m, 0, 0], dtype=np.uint64)
        coeffs = rand_elems(11, ncoef, W)
        d_coeffs = torch.from_numpy(coeffs.view(np.int64).copy())
        full = oracle.additive_fft(coeffs, basis, shift)
        lo, per = idist.shard_range(1 << m, rank, world)

        mine = idist.sharded_lde(lib, torch, d_coeffs, ncoef, basis, shift, rank, world)
        ok_lde = np.array_equal(mine.numpy().view(np.uint64), full[lo:lo + per])

        other = oracle.additive_fft(rand_elems(12, ncoef, W), basis, shift)
        d_other = torch.from_numpy(other[lo:lo + per].view(np.int64).copy())
        root, _ = idist.sharded_merkle_root(lib, torch, dist, [mine, d_other], per, cs, rank, world)
        ok_root = root == bytes(oracle.merkle_build([full, other], cs, True)[0])

        x = rand_elems(13, 1, W)[0]
        nxt = idist.sharded_fri_fold(lib, torch, mine, basis, shift, 4, x, rank, world)
        exp = oracle.fri_fold_additive(full, basis, shift, 4, x)
        ok_fold = np.array_equal(nxt.numpy().view(np.uint64), exp[lo // 4:(lo + per) // 4])
        ret[rank] = (ok_lde, ok_root, ok_fold)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_pipeline():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == (True, True, True), (r, ret[r])


def _fft_worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from emu_lib import emu
        from helpers import rand_elems
        from libiop_amd import dist as idist
        lib = emu()
        ok = []
        for m, kind in ((6, "std"), (8, "general")):
            basis = oracle.standard_basis(m, W) if kind == "std" else rand_elems(70 + m, m, W)
            shift = rand_elems(71 + m, 1, W)[0]
            coeffs = rand_elems(72 + m, 1 << m, W)
            full = oracle.additive_fft(coeffs, basis, shift)
            lo, per = idist.shard_range(1 << m, rank, world)
            plan = idist.DistributedFFTPlan(lib, torch, basis, shift, rank, world, torch.device("cpu"))
            mine = idist.distributed_fft(lib, torch, dist, plan, torch.from_numpy(coeffs[lo:lo + per].view(np.int64).copy()))
            ok.append(bool(np.array_equal(mine.numpy().view(np.uint64), full[lo:lo + per])))
            # the inverse direction (IFFT_over_field_subset on a full codeword, fft.tcc:126-204, 421-433): back to the coefficients, and
            # independent evaluations against the oracle's additive_IFFT
            back = idist.distributed_ifft(lib, torch, dist, plan, mine)
            ok.append(bool(np.array_equal(back.numpy().view(np.uint64), coeffs[lo:lo + per])))
            evals = rand_elems(73 + m, 1 << m, W)
            inv = idist.distributed_ifft(lib, torch, dist, plan, torch.from_numpy(evals[lo:lo + per].view(np.int64).copy()))
            ok.append(bool(np.array_equal(inv.numpy().view(np.uint64), oracle.additive_ifft(evals, basis, shift)[lo:lo + per])))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_distributed_full_size_fft(world):
    """One transform as long as its domain, forward and inverse, sharded by contiguous blocks: all-to-all transpose + peer exchanges."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_fft_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == [True] * 6, (r, ret[r])


def _mul_worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import libiop_amd as la
        import oracle
        from emu_lib import emu
        from libiop_amd import dist as idist
        lib = emu()
        P = la.EDWARDS_FR_MODULUS
        log_n, ncoef, cs = 9, 60, 2
        n = 1 << log_n
        shift_int = la.EDWARDS_FR_GENERATOR
        gen_int = pow(la.EDWARDS_FR_GENERATOR, (P - 1) >> log_n, P)
        shift_w = la.edwards_to_montgomery([shift_int])[0]
        rng = np.random.default_rng(5)
        cols = [la.edwards_to_montgomery([int.from_bytes(rng.bytes(32), "little") % P for _ in range(ncoef)]) for _ in range(2)]
        full = [oracle.multiplicative_fft(c, n, shift_w) for c in cols]
        # residue-sharded low-degree extension: the rank's positions are rank, rank + world, ...
        mine = [idist.sharded_mul_lde(lib, torch, la, torch.from_numpy(c.view(np.int64).copy()), ncoef, log_n, gen_int, shift_int, rank, world) for c in cols]
        ok_lde = all(np.array_equal(mine[k].numpy().view(np.uint64), full[k][rank::world]) for k in range(2))
        # Merkle root over both oracles, cosets of 2 and of 4
        ok_root = []
        for c in (2, 4):
            root, nodes = idist.sharded_mul_merkle_root(lib, torch, dist, la, mine, n // world, c, rank, world)
            want = oracle.merkle_build(full, c, False)
            L, per = n // c, n // c // world
            ok_root.append(root == bytes(want[0]) and np.array_equal(nodes[per - 1:].numpy(), want[L - 1 + rank * per:L - 1 + (rank + 1) * per]))
        # fold, then one more fold of the result (the residue sharding carries over to the next domain)
        x = la.edwards_to_montgomery([int.from_bytes(rng.bytes(32), "little") % P])[0]
        nxt = idist.sharded_mul_fri_fold(lib, torch, la, mine[0], log_n, gen_int, shift_int, 4, x, rank, world)
        exp = oracle.fri_fold_multiplicative(full[0], shift_w, 4, x)
        ok_fold = np.array_equal(nxt.numpy().view(np.uint64), exp[rank::world])
        sh2, g2 = pow(shift_int, 4, P), pow(gen_int, 4, P)
        nxt2 = idist.sharded_mul_fri_fold(lib, torch, la, nxt, log_n - 2, g2, sh2, 2, x, rank, world)
        exp2 = oracle.fri_fold_multiplicative(exp, la.edwards_to_montgomery([sh2])[0], 2, x)
        ok_fold2 = np.array_equal(nxt2.numpy().view(np.uint64), exp2[rank::world])
        allv = idist.gather_residues(torch, dist, nxt2, world)
        ok_gather = np.array_equal(allv.numpy().view(np.uint64), exp2)
        ret[rank] = (ok_lde, ok_root, ok_fold, ok_fold2, ok_gather)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_residue_sharded_multiplicative_pipeline(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_mul_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == (True, [True, True], True, True, True), (r, ret[r])


def _commit_worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import libiop_amd as la
        import oracle
        from emu_lib import emu
        from helpers import rand_elems
        from libiop_amd import dist as idist, fri, host
        lib = emu()
        htorch, to_dev = torch, (lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy()))
        # additive: contiguous blocks
        m, rs = 10, 2
        basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
        cw = oracle.additive_fft(rand_elems(3, 1 << (m - rs), W), basis, shift)
        loc = host.localization_parameter_to_array(2, m, rs)
        single = fri.fri_commit(lib, htorch, to_dev(cw), basis, shift, loc, 4)
        lo, per = idist.shard_range(1 << m, rank, world)
        roots, final = idist.sharded_fri_commit(lib, torch, dist, torch.from_numpy(cw[lo:lo + per].view(np.int64).copy()), basis, shift, loc, 4, rank, world)
        ok_add = roots == single.roots and np.array_equal(final, single.final_polynomial)
        # multiplicative: residue classes
        P = la.EDWARDS_FR_MODULUS
        log_n = 10
        shift_int, gen_int = la.EDWARDS_FR_GENERATOR, pow(la.EDWARDS_FR_GENERATOR, (P - 1) >> log_n, P)
        rng = np.random.default_rng(9)
        coeffs = la.edwards_to_montgomery([int.from_bytes(rng.bytes(32), "little") % P for _ in range(1 << (log_n - rs))])
        cwm = oracle.multiplicative_fft(coeffs, 1 << log_n, la.edwards_to_montgomery([shift_int])[0])
        locm = host.localization_parameter_to_array(2, log_n, rs)
        singlem = fri.fri_commit_multiplicative(lib, htorch, to_dev(cwm), log_n, shift_int, locm, 4)
        rootsm, finalm = idist.sharded_mul_fri_commit(lib, torch, dist, la, fri, torch.from_numpy(cwm[rank::world].view(np.int64).copy()),
                                                      log_n, gen_int, shift_int, locm, 4, rank, world)
        ok_mul = rootsm == singlem.roots and np.array_equal(finalm, singlem.final_polynomial)
        ret[rank] = (ok_add, ok_mul)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_fri_commit_equals_single_process(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_commit_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == (True, True), (r, ret[r])


# ---- the whole Aurora prover block-distributed over the ranks (libiop_amd/dist.py ShardedDeviceOps) ----
def _aurora_worker(rank, world, port, ret, log_n, num_inputs, rs_extra=5):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_lib import emu
        from libiop_amd import aurora, domains, r1cs
        from libiop_amd import dist as idist
        field = domains.GF192()
        ops = idist.ShardedDeviceOps(emu(), torch, torch.device("cpu"), field, idist.AuroraShard(dist, rank, world))
        n = 1 << log_n
        cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, 0x2204)
        params = aurora.AuroraParameters(field, n, n - 1, num_inputs, RS_extra_dimensions=rs_extra)
        d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
        transcript = idist.sharded_aurora_snark_prover(ops, cs, primary, params, d_z)
        ret[rank] = transcript.serialize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 7), (4, 8), (2, 9), (8, 7)])       # (8, 7): config 4's own split — 8 ranks x 4 cosets of every codeword
def test_sharded_aurora_prover_equals_oracle(world, log_n):
    """Every rank returns the transcript of the single-process oracle prover, byte for byte."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_aurora_worker, args=(world, port, ret, log_n, 15), nprocs=world, join=True)
    ref = oracle.aurora_prove(oracle.FIELD_GF192, log_n, 15, 0x2204)
    for r in range(world):
        assert ret[r] == ref, "rank %d" % r


def test_sharded_aurora_large_final_domain():
    """RS_extra_dimensions = 8: the last FRI domain (512 points) is large enough to stay distributed by size alone; it carries no
    oracle and must be gathered for the final interpolation."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_aurora_worker, args=(2, port, ret, 7, 15, 8), nprocs=2, join=True)
    ref = oracle.aurora_prove(oracle.FIELD_GF192, 7, 15, 0x2204, rs_extra=8)
    for r in range(2):
        assert ret[r] == ref, "rank %d" % r


# ---- the Fractal indexer and prover block-distributed over the ranks (subspace domains) ----
def _fractal_worker(rank, world, port, ret, log_n, num_inputs):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_lib import emu
        from libiop_amd import aurora, domains, fractal, r1cs
        from libiop_amd import dist as idist
        field = domains.GF192()
        ops = idist.ShardedDeviceOps(emu(), torch, torch.device("cpu"), field, idist.AuroraShard(dist, rank, world))
        n = 1 << log_n
        cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, 0x2205)
        params = fractal.FractalParameters(field, cs)
        index, (roots, _) = idist.sharded_fractal_snark_indexer(ops, cs, params)
        d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
        transcript = idist.sharded_fractal_snark_prover(ops, index, cs, primary, params, d_z)
        ret[rank] = (transcript.serialize(), [bytes(r) for r in roots])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 6), (4, 7), (8, 7)])
def test_sharded_fractal_prover_equals_oracle(world, log_n):
    """Every rank returns the index root and the transcript of the single-process oracle indexer / prover, byte for byte."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_fractal_worker, args=(world, port, ret, log_n, 15), nprocs=world, join=True)
    ref, ref_roots = oracle.fractal_prove(oracle.FIELD_GF192, log_n, 15, 0x2205)
    for r in range(world):
        assert ret[r][1] == ref_roots, "rank %d index root" % r
        assert ret[r][0] == ref, "rank %d" % r


# ---- multiplicative cosets: the provers distributed by residue class (libiop_amd/dist.py ResidueShardedDeviceOps) ----
def _residue_worker(rank, world, port, ret, protocol, log_n, num_inputs):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_lib import emu
        from libiop_amd import aurora, domains, fractal, r1cs
        from libiop_amd import dist as idist
        field = domains.EdwardsFr()
        ops = idist.sharded_ops(emu(), torch, torch.device("cpu"), field, idist.AuroraShard(dist, rank, world))
        assert isinstance(ops, idist.ResidueShardedDeviceOps)
        n = 1 << log_n
        if protocol == "aurora":
            cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, 0x2204)
            params = aurora.AuroraParameters(field, n, n - 1, num_inputs)
            d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
            ret[rank] = (idist.sharded_aurora_snark_prover(ops, cs, primary, params, d_z).serialize(), [])
        else:
            cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, 0x2205)
            params = fractal.FractalParameters(field, cs)
            index, (roots, _) = idist.sharded_fractal_snark_indexer(ops, cs, params)
            d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
            ret[rank] = (idist.sharded_fractal_snark_prover(ops, index, cs, primary, params, d_z).serialize(), [bytes(r) for r in roots])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("protocol,world,log_n,num_inputs", [("aurora", 2, 8, 15), ("aurora", 4, 9, 15), ("fractal", 2, 7, 0), ("fractal", 4, 9, 0),
                                                              ("fractal", 2, 8, 15)])
def test_residue_sharded_provers_equal_oracle(protocol, world, log_n, num_inputs):
    """Configs 1 and 5's field: every rank returns the transcript (and index root) of the single-process oracle prover, byte for byte."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_residue_worker, args=(world, port, ret, protocol, log_n, num_inputs), nprocs=world, join=True)
    if protocol == "aurora":
        ref, ref_roots = oracle.aurora_prove(oracle.FIELD_EDWARDS, log_n, num_inputs, 0x2204), []
    else:
        ref, ref_roots = oracle.fractal_prove(oracle.FIELD_EDWARDS, log_n, num_inputs, 0x2205)
    for r in range(world):
        assert ret[r][1] == ref_roots, "rank %d index root" % r
        assert ret[r][0] == ref, "rank %d" % r


def test_membership_proof_node_indices_match_oracle():
    import oracle
    from libiop_amd import dist as idist
    rng = np.random.default_rng(3)
    for num_leaves in (2, 8, 64, 1024):
        for _ in range(20):
            pos = sorted(set(int(p) for p in rng.integers(0, num_leaves, size=int(rng.integers(1, 12)))))
            assert idist.membership_proof_node_indices(num_leaves, pos) == [int(v) for v in oracle.membership_proof_indices(num_leaves, pos)]


# ---- the NATIVE provers distributed over the ranks (libiop_amd/cpp/dist.hpp behind iopx_aurora_prove_dist / iopx_fractal_*_dist): the C++
# prover inside the (CPU-compiled) library, its collectives forwarded to the gloo group through iopx_comm_create_callbacks ----
def _native_worker(rank, world, port, ret, protocol, field_code, log_n, num_inputs, seed, rs_extra):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_lib import emu
        lib = emu()
        comm = lib.comm_create_torch_callbacks(dist, rank, world)
        n = 1 << log_n
        inst = lib.aurora_example_instance(field_code, n, num_inputs, n - 1, seed)
        try:
            lib.comm_stats(reset=True)
            if protocol == "fri":                          # log_n = the codeword domain dimension; the seeded polynomial of degree 2^(dim - rs_extra)
                from libiop_amd import domains, r1cs
                f = domains.GF192() if field_code == 0 else domains.EdwardsFr()
                coeffs = np.ascontiguousarray(r1cs.seeded_elements(f, seed, 1 << (log_n - rs_extra)), dtype=np.uint64)
                d = lib.malloc(coeffs.nbytes)
                lib.h2d(d, coeffs)
                t = lib.fri_snark_prove(field_code, d, coeffs.shape[0], log_n, rs_extra, 2, 1, num_inputs, comm=comm)      # num_inputs carries the query repetitions
                lib.free(d)
                roots = []
            elif protocol == "aurora":
                t = lib.aurora_prove_dist(inst, comm, 128, rs_extra, 2)
                roots = []
            else:
                roots = lib.fractal_index_dist(inst, comm, 128, rs_extra, 2)
                t = lib.fractal_prove_dist(inst, comm, 128, rs_extra, 2)
            ret[rank] = (t, roots, lib.comm_stats())
        finally:
            lib.aurora_instance_free(inst)
            lib.comm_destroy(comm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,rs_extra", [(1, 6, 5), (2, 7, 5), (4, 8, 5), (8, 7, 5), (2, 7, 8)])
def test_native_sharded_aurora_prover_equals_oracle(world, log_n, rs_extra):
    """GF(2^192), contiguous cosets; (8, 7): config 4's own split; rs_extra 8: the last FRI domain is large enough to stay distributed by size."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_native_worker, args=(world, _free_port(), ret, "aurora", 0, log_n, 15, 0x2204, rs_extra), nprocs=world, join=True)
    ref = oracle.aurora_prove(oracle.FIELD_GF192, log_n, 15, 0x2204, rs_extra=rs_extra)
    for r in range(world):
        assert ret[r][0] == ref, "rank %d" % r
        assert ret[r][2][0] > 0, "no collective was issued"
        # round 5: the whole query phase is ONE all-reduce (answers and authentication paths of every tree in one arena) — it was two per tree
        if (world, log_n, rs_extra) in ((2, 7, 5), (4, 8, 5)):
            assert ret[r][2][0] <= 11, ret[r][2]


@pytest.mark.parametrize("world,log_n", [(2, 9), (4, 10), (8, 9)])
def test_native_sharded_aurora_prover_over_the_prime_field_equals_oracle(world, log_n):
    """181-bit field, residue classes (leaf digests exchanged by all-to-all)."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_native_worker, args=(world, _free_port(), ret, "aurora", 1, log_n, 15, 0x2204, 5), nprocs=world, join=True)
    ref = oracle.aurora_prove(oracle.FIELD_EDWARDS, log_n, 15, 0x2204)
    for r in range(world):
        assert ret[r][0] == ref, "rank %d" % r


@pytest.mark.parametrize("world,field_code,log_n,num_inputs", [(2, 0, 6, 15), (4, 0, 7, 15), (8, 0, 7, 15), (2, 1, 7, 0), (4, 1, 8, 15), (8, 1, 8, 0)])
def test_native_sharded_fractal_prover_equals_oracle(world, field_code, log_n, num_inputs):
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_native_worker, args=(world, _free_port(), ret, "fractal", field_code, log_n, num_inputs, 0x2205, 3), nprocs=world, join=True)
    ref, ref_roots = oracle.fractal_prove(oracle.FIELD_GF192 if field_code == 0 else oracle.FIELD_EDWARDS, log_n, num_inputs, 0x2205)
    for r in range(world):
        assert ret[r][1] == ref_roots, "rank %d index root" % r
        assert ret[r][0] == ref, "rank %d" % r
        if (world, field_code, log_n) == (4, 1, 8):
            assert ret[r][2][0] <= 20, ret[r][2]                  # 33 before the query phase's collectives were merged into one


def _bad_witness_instance(lib, field_code, log_n, k, seed):
    """The seeded constraint system with one auxiliary variable changed (Az * Bz != Cz): an instance handle built through iopx_aurora_instance_create."""
    import torch
    import head_cases as hc
    from libiop_amd import domains, r1cs
    field = domains.GF192() if field_code == 0 else domains.EdwardsFr()
    ops = domains.DeviceOps(lib, torch, torch.device("cpu"), field)
    n = 1 << log_n
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, k, n - 1, seed)
    z = np.concatenate([np.asarray(primary, dtype=np.uint64).reshape(-1, 3), np.asarray(auxiliary, dtype=np.uint64).reshape(-1, 3)])
    z[k + 5] = z[k + 6]
    return lib.aurora_instance(field_code, [hc.csr(ops, M) for M in (cs.A, cs.B, cs.C)], n - 1, k, z)


def _bad_witness_worker(rank, world, port, ret, field_code, log_n):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_lib import emu
        lib = emu()
        comm = lib.comm_create_torch_callbacks(dist, rank, world)
        inst = _bad_witness_instance(lib, field_code, log_n, 15, 0x2204)
        try:
            lib.profile_begin()
            t = lib.aurora_prove_dist(inst, comm, 128, 5, 2)
            prof = lib.profile_report()
            ret[rank] = (t, sum(v[0] for k, v in prof.items() if k.startswith("k_ldt_combine")))
        finally:
            lib.aurora_instance_free(inst)
            lib.comm_destroy(comm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,field_code,log_n", [(2, 0, 8), (4, 0, 9), (2, 1, 9)])
def test_native_sharded_unsatisfied_witness_takes_the_reference_schedule_on_every_rank(world, field_code, log_n, monkeypatch):
    """An unsatisfied witness over the ranks: the confirmation window's mismatch count is all-reduced, so EVERY rank discards the head schedule's f_1 and
    proves by the reference's schedule — the bytes of the single-process prover run with IOPX_HEAD_EVAL=0."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from emu_lib import emu
    lib = emu()
    monkeypatch.setenv("IOPX_HEAD_EVAL", "0")
    inst = _bad_witness_instance(lib, field_code, log_n, 15, 0x2204)
    try:
        expected = lib.aurora_prove(inst)
    finally:
        lib.aurora_instance_free(inst)
    monkeypatch.delenv("IOPX_HEAD_EVAL")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bad_witness_worker, args=(world, _free_port(), ret, field_code, log_n), nprocs=world, join=True)
    for r in range(world):
        assert ret[r][0] == expected, "rank %d" % r
    # the whole-domain combination ran on every rank after the head (rank 0) and the confirmation window (its owner)
    assert all(ret[r][1] >= 1 for r in range(world)) and sum(ret[r][1] for r in range(world)) == world + 2, dict((r, ret[r][1]) for r in range(world))


# ---- phase 1 of the replicated transforms split over the ranks (iopx_comm_bind_transforms; fft_add.hip run_phase1) ----
P1_SHARD_ENV = {"IOPX_P1_SHARD_MIN_D": "6", "IOPX_TILE_BITS": "5", "IOPX_P1_COLS": "2", "IOPX_P2_COLS": "2", "IOPX_P2_TOP": "2"}


def _phase1_worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        import oracle
        from emu_lib import emu
        from helpers import rand_elems
        lib = emu()
        comm = lib.comm_create_torch_callbacks(dist, rank, world)
        lib.c.iopx_comm_bind_transforms.argtypes = [ctypes.c_void_p]
        ok, calls = [], []
        for m, kind in ((8, "std"), (9, "general"), (11, "std"), (5, "std")):         # m = 5: below the threshold, stays whole
            basis = oracle.standard_basis(m, W) if kind == "std" else rand_elems(70 + m, m, W)
            shift = rand_elems(71 + m, 1, W)[0]
            coeffs = rand_elems(72 + m, 1 << m, W)
            full = oracle.additive_fft(coeffs, basis, shift)
            lib.comm_stats(reset=True)
            lib._check(lib.c.iopx_comm_bind_transforms(comm))
            try:
                got = lib.additive_FFT(coeffs, basis, shift)
                back = lib.additive_IFFT(full, basis, shift)
                short = coeffs[: (1 << (m - 2)) - 3]                                  # a low-degree extension: phase 1 on 2^(m-2) coefficients
                lde = lib.additive_FFT(short, basis, shift)
            finally:
                lib._check(lib.c.iopx_comm_bind_transforms(None))
            calls.append(lib.comm_stats()[0])
            ok.append(bool(np.array_equal(got, full) and np.array_equal(back, coeffs) and np.array_equal(lde, oracle.additive_fft(short, basis, shift))))
        ret[rank] = (ok, calls)
        lib.comm_destroy(comm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_phase1_split_over_the_ranks(world, monkeypatch):
    """Forward and inverse transforms with their phase 1 split by residue class of the coefficient index (levels < log2 N on the whole vector,
    the rest on the rank's class, one all-gather): results equal the oracle's on every rank, and the collectives were really issued."""
    for k, v in P1_SHARD_ENV.items():
        monkeypatch.setenv(k, v)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_phase1_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        ok, calls = ret[r]
        assert ok == [True] * 4, (r, ok)
        assert calls[0] >= 2 and calls[1] >= 2 and calls[2] >= 3 and calls[3] == 0, (r, calls)


@pytest.mark.parametrize("world,protocol,field_code,log_n", [(2, "aurora", 0, 7), (8, "aurora", 0, 9), (4, "fractal", 0, 7)])
def test_native_sharded_provers_with_phase1_split(world, protocol, field_code, log_n, monkeypatch):
    """The native distributed provers with the phase-1 split active at test sizes (d >= 16 by default).  Aurora's default schedule re-extends its
    codewords without a coefficient form, so no replicated basis conversion is left to split: its cases run the reference's schedule
    (IOPX_HEAD_EVAL=0), which converts every oracle; Fractal keeps replicated conversions either way."""
    import oracle
    mgr = mp.Manager()
    plain, ret = mgr.dict(), mgr.dict()
    seed = 0x2204 if protocol == "aurora" else 0x2205
    args = (protocol, field_code, log_n, 15, seed, 5 if protocol == "aurora" else 3)
    if protocol == "aurora":
        monkeypatch.setenv("IOPX_HEAD_EVAL", "0")
    mp.spawn(_native_worker, args=(world, _free_port(), plain) + args, nprocs=world, join=True)       # default tuning: transforms stay whole at this size
    for k, v in P1_SHARD_ENV.items():
        monkeypatch.setenv(k, v)
    mp.spawn(_native_worker, args=(world, _free_port(), ret) + args, nprocs=world, join=True)
    if protocol == "aurora":
        ref, ref_roots = oracle.aurora_prove(oracle.FIELD_GF192, log_n, 15, seed), []
    else:
        ref, ref_roots = oracle.fractal_prove(oracle.FIELD_GF192, log_n, 15, seed)
    for r in range(world):
        assert ret[r][0] == ref and ret[r][1] == ref_roots, "rank %d" % r
        assert plain[r][0] == ref
        assert ret[r][2][0] > plain[r][2][0], "the transforms issued no collective: %r vs %r" % (ret[r][2], plain[r][2])


@pytest.mark.parametrize("world,field_code,dim", [(2, 0, 10), (4, 0, 11), (2, 1, 12)])
def test_native_sharded_fri_snark_equals_oracle(world, field_code, dim):
    """BASELINE config 3's prover (iopx_fri_snark_prove_dist) distributed over the ranks."""
    import oracle
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_native_worker, args=(world, _free_port(), ret, "fri", field_code, dim, 8, 5, 2), nprocs=world, join=True)
    ref = oracle.fri_snark_prove(oracle.FIELD_GF192 if field_code == 0 else oracle.FIELD_EDWARDS, dim, 2, 2, 1, 8, 5)
    for r in range(world):
        assert ret[r][0] == ref, "rank %d" % r
        assert ret[r][2][0] > 0


# ---- ONE transform as long as its domain across the ranks, natively (libiop_amd/csrc/fft_add_dist.hip behind iopx_add_[i]fft_gf192_dist_dev) ----
def _native_fft_worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from emu_lib import emu
        from helpers import rand_elems
        lib = emu()
        comm = lib.comm_create_torch_callbacks(dist, rank, world)
        ok = []
        for m, kind in ((6, "std"), (8, "general"), (11, "aurora")):
            if kind == "std":
                basis, shift = oracle.standard_basis(m, W), np.zeros(W, dtype=np.uint64)
            elif kind == "aurora":
                basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
            else:
                basis, shift = rand_elems(70 + m, m, W), rand_elems(71 + m, 1, W)[0]
            coeffs = rand_elems(72 + m, 1 << m, W)
            full = oracle.additive_fft(coeffs, basis, shift)
            per = (1 << m) // world
            lo = rank * per
            d_in, d_out, d_back = lib.malloc(per * 24), lib.malloc(per * 24), lib.malloc(per * 24)
            try:
                lib.h2d(d_in, np.ascontiguousarray(coeffs[lo:lo + per]))
                lib.additive_FFT_dist_dev(comm, d_in, basis, shift, d_out)
                got = np.empty((per, W), dtype=np.uint64)
                lib.d2h(got, d_out)
                ok.append(bool(np.array_equal(got, full[lo:lo + per])))
                lib.additive_FFT_dist_dev(comm, d_out, basis, shift, d_back, inverse=True)           # back to the coefficients
                lib.d2h(got, d_back)
                ok.append(bool(np.array_equal(got, coeffs[lo:lo + per])))
                evals = rand_elems(73 + m, 1 << m, W)                                               # independent evaluations against the oracle's IFFT
                lib.h2d(d_in, np.ascontiguousarray(evals[lo:lo + per]))
                lib.additive_FFT_dist_dev(comm, d_in, basis, shift, d_out, inverse=True)
                lib.d2h(got, d_out)
                ok.append(bool(np.array_equal(got, oracle.additive_ifft(evals, basis, shift)[lo:lo + per])))
            finally:
                for d in (d_in, d_out, d_back):
                    lib.free(d)
        before = lib.comm_stats()[0]
        if world > 2:                                        # m = 2 log2(world) - 1: nothing to put in a transpose chunk — refused, no collective issued
            m = 2 * (world.bit_length() - 1) - 1
            per = max(1, (1 << m) // world)
            d_a, d_b = lib.malloc(per * 24), lib.malloc(per * 24)
            for inverse in (False, True):
                try:
                    lib.additive_FFT_dist_dev(comm, d_a, oracle.standard_basis(m, W), np.zeros(W, dtype=np.uint64), d_b, inverse=inverse)
                    ok.append(False)
                except ValueError:
                    ok.append(True)
            lib.free(d_a)
            lib.free(d_b)
            ok.append(lib.comm_stats()[0] == before)
        ret[rank] = (ok, before)
        lib.comm_destroy(comm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_distributed_full_size_fft(world):
    """additive_FFT / additive_IFFT of a polynomial as long as its domain, block-distributed: all-to-all transpose, top levels by shard exchanges, local
    transform, cross-block butterflies — equal to the oracle's single-process transform on every rank."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_native_fft_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r][0] == [True] * (9 if world == 2 else 12), (r, ret[r])
        assert ret[r][1] > 0
