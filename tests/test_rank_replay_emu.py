"""The replay communicator (iopx_comm_create_replay): rank r of N played alone, collectives completed locally.  Checked on the CPU build of
the kernel sources: the distributed provers run to the end for every (world, rank) bench.py replays, issue the collectives the real
rank issues, and a one-rank replay IS the single-GPU prover (byte-equal to the oracle)."""
import pytest

import oracle
from emu_lib import emu

LOG_N = 8


def _instance(lib, field=0, k=15, seed=0x2204):
    n = 1 << LOG_N
    return lib.aurora_example_instance(field, n, k, n - 1, seed)


def test_one_rank_replay_is_the_single_gpu_prover():
    lib = emu()
    inst = _instance(lib)
    comm = lib.comm_create_replay(0, 1)
    try:
        assert lib.aurora_prove_dist(inst, comm) == oracle.aurora_prove(oracle.FIELD_GF192, LOG_N, 15, 0x2204)
    finally:
        lib.comm_destroy(comm)
        lib.aurora_instance_free(inst)


@pytest.mark.parametrize("world,rank", [(2, 0), (2, 1), (4, 0), (4, 3), (8, 0), (8, 7)])
def test_aurora_rank_replay_runs_and_counts_the_ranks_collectives(world, rank):
    lib = emu()
    inst = _instance(lib)
    comm = lib.comm_create_replay(rank, world)
    try:
        lib.comm_stats(reset=True)
        t = lib.aurora_prove_dist(inst, comm)
        calls, payload = lib.comm_stats()
        assert len(t) > 1000 and calls > 0 and payload > 0          # an argument came out (its bytes mean nothing) and the rank did talk
        # the schedule must be the real rank's: the head route (a fallback to the reference's schedule would show as a much larger LDT input)
        t2 = lib.aurora_prove_dist(inst, comm)
        assert len(t2) == len(t)
    finally:
        lib.comm_destroy(comm)
        lib.aurora_instance_free(inst)


@pytest.mark.parametrize("world,rank", [(2, 1), (4, 0)])
def test_fractal_rank_replay_runs(world, rank):
    lib = emu()
    n = 1 << LOG_N
    inst = lib.aurora_example_instance(1, n, 0, n - 1, 0x2205)
    comm = lib.comm_create_replay(rank, world)
    try:
        roots = lib.fractal_index_dist(inst, comm)
        assert len(roots) == 1
        assert len(lib.fractal_prove_dist(inst, comm)) > 1000
    finally:
        lib.comm_destroy(comm)
        lib.aurora_instance_free(inst)
