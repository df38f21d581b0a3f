#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X prover hot path (BASELINE.json metric: FFT field-ops/s).

Workload at N = 1 (BASELINE.json configs[1]): the standalone additive FFT over GF(2^192) of 2^22 random
coefficients on the standard-basis subspace of dimension 22, shift 0 (libiop/profiling/
instrument_algebra.cpp:84-94).  A "step" is one such transform with the coefficients already resident in HBM.
field-ops are counted with the REFERENCE's operation count for this size (1.5 n m
multiplications + (n/2) m (m-1)/2 + n m additions), whatever algorithm runs.

At N > 1 every rank transforms its own 2^22-coefficient polynomial (independent units, no data-path
collective; "weak" scaling); the value is the sum over ranks divided by the slowest rank's time.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 22
ELEM = 24
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def ref_field_ops(m):
    n = 1 << m
    mults = 3 * n * m // 2
    adds = (n // 2) * (m * (m - 1) // 2) + n * m
    return mults, adds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true",
                    help="N > 1: ONE 2^(log_n + log2 N)-point transform sharded across the ranks (libiop_amd/dist.py: all-to-all "
                         "transpose + peer exchanges) instead of one independent transform per rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import libiop_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    lib = libiop_amd.lib()
    lib.init(local_rank)
    stream = torch.cuda.current_stream()
    lib.set_stream(stream.cuda_stream)

    m = args.log_n
    n = 1 << m
    basis = libiop_amd.standard_basis(m)
    shift = np.zeros(3, dtype=np.uint64)
    rng = np.random.Generator(np.random.PCG64(0x2201 + rank))
    coeffs_h = rng.integers(0, 2**64, size=(n, 3), dtype=np.uint64)
    # device-resident buffers owned by torch (int64 storage = raw words)
    d_in = torch.from_numpy(coeffs_h.view(np.int64)).to(dev)
    d_out = torch.empty_like(d_in)

    def step():
        lib.additive_FFT_dev(d_in.data_ptr(), n, basis, shift, d_out.data_ptr())

    total_m = m
    if args.sharded and world > 1:
        from libiop_amd import dist as idist
        total_m = m + (world.bit_length() - 1)
        big_basis = libiop_amd.standard_basis(total_m)
        plan = idist.DistributedFFTPlan(lib, torch, big_basis, shift, rank, world, dev)

        def step():          # noqa: F811 — the rank's block of the coefficients in, its block of the evaluations out
            idist.distributed_fft(lib, torch, dist, plan, d_in)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel durations, live, with HIP events on the stream the kernels are launched on
    lib.profile_begin()
    for _ in range(args.steps):
        step()
    prof = lib.profile_report()
    dom_name, (dom_cnt, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    dom_avg_s = dom_ms / dom_cnt / 1e3
    # every FFT pass kernel sweeps the whole vector once: algorithmic bytes per launch = read + write of n elements
    alg_bytes = 2 * n * ELEM
    achieved = alg_bytes / dom_avg_s / 1e9

    # HBM traffic of the dominant kernel from the committed PMC run (profiles/), when it is for this kernel and size
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if tj.get("log_n") == m and dom_name in tj.get("kernels", {}):
            traffic = tj["kernels"][dom_name]["traffic_bytes_per_launch"]
    except (OSError, ValueError):
        pass

    mults, adds = ref_field_ops(total_m)
    ms_per_step = dt / args.steps * 1e3
    units = 1 if (args.sharded and world > 1) else world       # one big transform, or one transform per rank
    value = units * (mults + adds) / (dt / args.steps)

    out = {
        "metric": "fft_field_ops_per_s",
        "value": value,
        "unit": "field-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "gf2^192 (u32 VALU bit-ops)",
        "data": "synthetic",
        "config": {"workload": "additive FFT over GF(2^192), 2^%d coefficients -> 2^%d-point standard-basis subspace, shift 0 "
                               "(BASELINE configs[1]); one transform per GPU" % (m, m),
                   "log_n": m, "field": "gf192", "ref_mults_per_step": mults, "ref_adds_per_step": adds,
                   "field_mults_per_s": units * mults / (dt / args.steps),
                   "multi_gpu": ("one 2^%d-point transform sharded over %d ranks (all-to-all transpose + peer exchanges)" % (total_m, world))
                   if (args.sharded and world > 1) else "one independent transform per rank, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": dom_name, "launches_per_step": dom_cnt / args.steps, "avg_launch_ms": dom_avg_s * 1e3,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "gfx950 has no carry-less multiply: the GF(2^192) butterflies are integer-ALU-bound, see DESIGN.md",
                     "kernels_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
                     # the bound that actually binds: GF(2^192) products per second of the dominant kernel (each launch that
                     # twists multiplies every element once) against the measured rate of the in-register multiplier
                     # (tools/ubench/mul_rates.hip: 4.4e10/s general operands)
                     "valu": {"products_per_s": (n * (m if dom_name.startswith("k_phase1") else m / 2)) / (dom_ms / args.steps / 1e3),
                              "multiplier_peak_per_s": 4.4e10,
                              "frac": (n * (m if dom_name.startswith("k_phase1") else m / 2)) / (dom_ms / args.steps / 1e3) / 4.4e10}},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        t0 = time.perf_counter()
        ref = oracle.additive_fft(coeffs_h, basis, shift)
        cpu_s = time.perf_counter() - t0
        got = d_out.cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ref), "GPU output differs from the CPU oracle"
        out["cpu_baseline"] = {"value": (mults + adds) / cpu_s, "unit": "field-ops/s", "cores": 1, "kind": "port",
                               "sample": "the full workload once (2^%d-point additive FFT, PCLMUL gf192, 1 thread): %.2f s; "
                                         "output compared bit-for-bit with the GPU result" % (m, cpu_s),
                               "seconds": cpu_s}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
