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
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_full_size_fft(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_fft_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == [True, True], (r, ret[r])
