#!/usr/bin/env python3
"""Soak test of the native Aurora prover on cuda:0: 301 proofs of the 2^20 instance, every transcript equal to the first, free HBM printed every
100 proofs (a leak in the buffer pool or the pinned staging would show as a falling figure)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libiop_amd
lib = libiop_amd.Library()
comm = None
if "RANK" in os.environ:          # under `python -m torch.distributed.run --nproc-per-node N tools/soak.py`: the distributed code path over an RCCL communicator
    import torch.distributed as dist
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    lib.init(local_rank)
    comm = lib.comm_create_rccl_from_torch(dist, rank, world, torch.device("cuda", local_rank))
lib.set_stream(torch.cuda.current_stream().cuda_stream)
inst = lib.aurora_example_instance(0, 1 << 20, 15, (1 << 20) - 1, 0x20)
first = None
t0 = time.time()
for i in range(301):
    t = lib.aurora_prove_dist(inst, comm, 128, 5, 2) if comm is not None else lib.aurora_prove(inst, 128, 5, 2)
    if first is None:
        first = bytes(t)
    assert bytes(t) == first
    if i % 100 == 0:
        free, total = torch.cuda.mem_get_info()
        print(i, "free GB %.3f" % (free / 2**30), "elapsed %.1f s" % (time.time() - t0), flush=True)
lib.aurora_instance_free(inst)
if comm is not None:
    lib.comm_destroy(comm)
    dist.destroy_process_group()
print("ok", len(first))
