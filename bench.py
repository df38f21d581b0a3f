#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X prover hot path.

BASELINE.json metric: "prover sec + FFT field-ops/s, Aurora 2^20 R1CS, 1/2/4/8 MI355X vs CPU".

Workload (BASELINE.json configs[3], run on as many GPUs as given): the Aurora SNARK prover for the synthetic
2^20-constraint R1CS over GF(2^192) (generate_r1cs_example(2^20, 15, 2^20 - 1), profiling/instrument_aurora_snark.cpp:
108-110; security 128, RS_extra_dimensions 5, FRI localization 2, non-zk, BLAKE2b).  A "step" is one complete proof —
witness -> codewords -> lincheck / sumcheck -> LDT reducer -> FRI -> proof of work -> transcript — with the instance and
the witness already resident in HBM (libiop_amd/aurora.py over the C ABI).

`value` = the proof's FFT work in the REFERENCE's operation count (per 2^m-point transform 1.5 n m
multiplications + (n/2) m (m-1)/2 + n m additions, summed over every transform the reference prover runs) divided by the
prover's wall-clock seconds: prover seconds and FFT field-ops/s in one number; `ms_per_step` is the prover time itself.
The roofline line is for the dominant kernel of THAT run; config.secondary holds configs[1] (one 2^22 FFT) and
config.secondary_fractal configs[4]'s Fractal prover on this GPU.

N > 1 (one process per GPU): the SAME native prover, iopx_aurora_prove_dist, with every codeword-domain vector split over the ranks by
contiguous cosets (libiop_amd/cpp/dist.hpp) and an RCCL communicator created through the C ABI; "strong" scaling: one proof.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N ...                      (starts torch.distributed.run itself, as a child process, when RANK is unset)
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

ELEM = 24


def kernel_sources_digest():
    """sha256 over the kernel and prover sources: profile-derived figures are only quoted when they were collected on these sources."""
    import hashlib
    h = hashlib.sha256()
    for sub in ("csrc", "cpp"):                 # the kernels and the prover that decides which of them run, over what
        for dp, _, fs in sorted(os.walk(os.path.join(ROOT, "libiop_amd", sub))):
            for f in sorted(fs):
                if f.endswith((".hip", ".h", ".hpp")):
                    h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 0x2204


def ref_fft_ops(m):
    """(multiplications, additions) the reference's additive FFT / IFFT performs on a 2^m-point domain."""
    n = 1 << m
    return 3 * n * m // 2, (n // 2) * (m * (m - 1) // 2) + n * m


def aurora_transform_inventory(log_n, rs_extra, final_dim):
    """Every transform of the reference's non-zk Aurora prover as (what, domain dimension) — SURVEY.md §8d, confirmed there by
    the size trace of the reference prover: 8 codeword FFTs, the interpolations over the 2^log_n domains, one 2^(log_n+1)
    known-degree IFFT, the small input-domain and FRI-final transforms."""
    L = log_n + rs_extra
    inv = [("FFT f_w / f_Az / f_Bz / f_Cz / f_1v / p_alpha_prime / p_alpha_ABC / h -> codeword domain", L)] * 8
    inv += [("IFFT f_w' over the variable domain", log_n), ("FFT f_1v over the variable domain", log_n)]
    inv += [("IFFT Az / Bz / Cz over the constraint domain", log_n)] * 3
    inv += [("IFFT p_alpha over the summation domain", log_n)] * 2
    inv += [("IFFT_of_known_degree of the sumcheck polynomial", log_n + 1)]
    inv += [("IFFT f_1v over the input domain", 4)] * 2 + [("IFFT of the last FRI codeword", final_dim)]
    return inv


def launcher_command(gpus, argv):
    """The command `python bench.py --gpus N` re-runs itself with: one rank per GPU of this node, rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def host_description():
    """CPU model and core count of the box the baseline runs on (lscpu / nproc)."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "logical_cores": os.cpu_count()}


def _oracle_prove(k):
    import oracle
    t0 = time.perf_counter()
    oracle.aurora_prove(oracle.FIELD_GF192, k, 15, SEED)
    return time.perf_counter() - t0


def cpu_baseline_leg(k, all_cores, reps=5):
    """Times the oracle prover (test infrastructure: used here as the reported CPU baseline only) on the 2^k-constraint sample: one warm-up run, then
    `reps` timed runs, the median reported."""
    import oracle
    oracle.aurora_prove(oracle.FIELD_GF192, k, 15, SEED)          # warm-up (page cache, tables)
    times, ref, blocks = [], None, None
    for _ in range(max(reps, 1)):
        oracle.block_times()                 # reset
        t0 = time.perf_counter()
        ref = oracle.aurora_prove(oracle.FIELD_GF192, k, 15, SEED)
        times.append(time.perf_counter() - t0)
        blocks = oracle.block_times()
    cpu_s = sorted(times)[len(times) // 2]
    out = {"transcript": ref, "seconds": cpu_s, "runs_s": [round(t, 4) for t in times], "log_n": k, "host": host_description(), "blocks": blocks}
    if all_cores:
        # not in the reference (it is single-threaded): one independent proof per logical core, all at once — a throughput figure
        import multiprocessing as mp
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        cores = min(usable, 16)          # bounded: the box's container does not necessarily get every logical core it can see
        kk = max(k - 1, 8)
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_oracle_prove, [kk] * cores)
        out["all_cores"] = {"cores": cores, "log_n": kk, "seconds_for_one_proof_per_core": time.perf_counter() - t0}
    return out


def alu_model():
    """profiles/r06_alu_model.json (tools/alu_model.py): the VALU-issue ceiling of the butterfly kernels from their ISA and the measured
    per-class instruction costs; None when the file is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r06_alu_model.json")))
    except (OSError, ValueError):
        return None


# The reference's own profiling blocks (libff::enter_block) and the device kernels that do their work: per-stage time on both legs under one
# vocabulary.  Device side = HIP-event time of the kernels inside one proof; CPU side = the oracle's timers under the same names (oracle/field.hpp).
REFERENCE_BLOCKS = [
    ("Call to additive_FFT_wrapper", "libiop/algebra/fft.tcc:210", ("k_bfly_upper_fwd", "k_bfly_edge_fwd", "k_phase1_fwd", "k_rs_combine", "k_pad_copy", "k_pow_direct", "k_pow_expand", "k_fill")),
    ("Call to additive_IFFT_wrapper", "libiop/algebra/fft.tcc:222", ("k_bfly_upper_inv", "k_bfly_edge_inv", "k_phase1_inv")),
    ("Construct Merkle tree", "libiop/bcs/bcs_prover.tcc:43", ("k_merkle_",)),
    ("evaluating next FRI codeword", "libiop/protocols/ldt/fri/fri_ldt.tcc:519", ("k_fri_fold",)),
    ("pow", "libiop/bcs/bcs_prover.tcc:52", ("k_pow_blake2b",)),
    ("Obtain transcript", "libiop/snark/aurora_snark.tcc:138", ("k_gather_nodes", "k_gather_responses")),
]


def device_stages(prof):
    """Kernel time of one proof (HIP events) grouped under the reference's block names; what no block of the reference times on its own — the
    virtual oracles' evaluated_contents, the sparse products, constant uploads — is 'other (virtual oracles, sparse products, uploads)'."""
    out, used = {}, set()
    for name, where, prefixes in REFERENCE_BLOCKS:
        members = [k for k in prof if k.startswith(prefixes)]
        used.update(members)
        out[name] = {"ms": round(sum(prof[k][1] for k in members), 4), "launches": sum(prof[k][0] for k in members), "reference_block": where}
    rest = [k for k in prof if k not in used]
    out["other (virtual oracles, sparse products, uploads)"] = {"ms": round(sum(prof[k][1] for k in rest), 4), "launches": sum(prof[k][0] for k in rest)}
    return out


def headline_cpu_figure():
    """The oracle prover at the HEADLINE size, measured once per round on the GPU box's host (tools/cpu_baseline_sizes.py, 40 minutes of one core: not repeated per
    run): profiles/r06_cpu_baseline_sizes.json (r04's when that is missing), with the transcript digest its time was accepted on — the digest the device
    prover's transcript must hash to."""
    for name in ("r06_cpu_baseline_sizes.json", "r04_cpu_baseline_sizes.json"):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", name)))
            e = next(x for x in j["sizes"] if x["log_n"] == 20)
            return {"log_n": 20, "prover_seconds": e["prover_seconds"], "field_ops_per_s": e["field_ops_per_s"], "cores": j["cores"], "kind": j["kind"],
                    "host": j["host"], "transcript_blake2b": e["transcript_blake2b"], "source": "profiles/%s (tools/cpu_baseline_sizes.py --log-n 20)" % name}
        except (OSError, ValueError, KeyError, StopIteration):
            continue
    return None


def reference_own_digest_equal(log_n, transcript):
    """True / False when tests/golden/reference_over_shim.json holds the digest libiop's own Aurora prover produced for this instance, else None."""
    if transcript is None:
        return None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "reference_over_shim.json")) as f:
            doc = json.load(f)
    except OSError:
        return None
    for e in doc.get("large_entries", []) + doc.get("entries", []):
        if e["protocol"] == "aurora" and e["field"] == "gf192" and e["log_n"] == log_n and e["num_inputs"] == 15 and e["seed"] == 0x2204:
            return __import__("hashlib").blake2b(transcript.serialize(), digest_size=32).hexdigest() == e["transcript_blake2b"]
    return None


def rank_replay(lib, torch, native, params, world, rank, steps=5, warmup=2):
    """One rank's compute path of the N-rank prover, measured on ONE GPU: iopx_aurora_prove_dist over a replay communicator
    (iopx_comm_create_replay: rank `rank` of `world` alone, every collective completed locally on the stream — all-gathers filled from this rank's
    own part, all-reduces and broadcasts left as they are).  Times the kernels the rank runs between its collectives; the collectives' own latency
    over xGMI is NOT in the figure (modelled in DESIGN.md section 6), and the transcript of such a proof is meaningless.  The proof-of-work grind of a
    replayed rank covers all ranks' candidate ranges (nobody to hear a hit from): pow_ms is reported, and ms_per_proof_pow_adjusted takes
    (world - 1) / world of it off again."""
    comm = lib.comm_create_replay(rank, world)
    try:
        for _ in range(warmup):
            lib.aurora_prove_dist(native, comm, 128, params.RS_extra_dimensions, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lib.aurora_prove_dist(native, comm, 128, params.RS_extra_dimensions, 2)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        lib.comm_stats(reset=True)
        lib.profile_begin()                           # while the profiler records, the library keeps side-stream sections on the main stream
        lib.aurora_prove_dist(native, comm, 128, params.RS_extra_dimensions, 2)
        prof = lib.profile_report()
        calls, payload = lib.comm_stats()
    finally:
        lib.comm_destroy(comm)
    pow_ms = sum(v[1] for k, v in prof.items() if k.startswith("k_pow_blake2b"))
    return {"world": world, "rank": rank, "ms_per_proof": round(ms, 3), "steps": steps,
            "pow_ms": round(pow_ms, 3), "ms_per_proof_pow_adjusted": round(ms - pow_ms * (world - 1) / world, 3),
            "kernels_ms_total": round(sum(v[1] for v in prof.values()), 3),
            "kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]},
            "collectives_per_proof": calls, "collective_payload_bytes_this_rank": payload}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of the number of R1CS constraints")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cross-check", action="store_true",
                    help="skip the second prover's proof after the timed loop (profiling runs: its reference-schedule launches would mix into the per-kernel counters)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the multi-GPU operator set even with one rank (exercises the RCCL calls on a 1-GPU box; run under torch.distributed.run)")
    ap.add_argument("--throughput", action="store_true",
                    help="with --gpus N: N independent proofs, one per rank, no collective (a labelled secondary figure: weak scaling; the default shards ONE proof over the ranks)")
    ap.add_argument("--cpu-log-n", type=int, default=11, help="size of the CPU-baseline sample (oracle prover); 2^11: about 4 s per run on one core, 2^12: 8 s")
    ap.add_argument("--cpu-reps", type=int, default=5, help="timed runs of the CPU-baseline sample after one warm-up run (the median is reported)")
    ap.add_argument("--replay-rank", type=int, default=None,
                    help="with --world N on ONE GPU: time rank R of the N-rank prover alone (collectives completed locally; compute path only) and print that line")
    ap.add_argument("--world", type=int, default=8, help="the world size --replay-rank plays a rank of")
    ap.add_argument("--no-warm", action="store_true", help="do not call iopx_aurora_instance_warm: config.first_proof_ms is then the cold first proof")
    ap.add_argument("--no-rank-replay", action="store_true", help="skip config.rank_replay (ranks 0 and N-1 of 2, 4, 8 played alone on this GPU)")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time one independent oracle proof per host core at once (not in the reference, which is single-threaded)")
    args = ap.parse_args()

    # RCCL prints a version banner on STDOUT at NCCL_DEBUG=VERSION/INFO (seen on the GPU box: five lines ahead of the JSON line);
    # stdout carries the one JSON line only, so the debug level is pinned to WARN unless IOPX_NCCL_DEBUG says otherwise
    os.environ["NCCL_DEBUG"] = os.environ.get("IOPX_NCCL_DEBUG", "WARN")
    if args.gpus > 1 and "RANK" not in os.environ:
        # Started as `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (torch is not even imported),
        # the ranks run as CHILD processes of torch.distributed.run and this process only forwards their exit code.
        import subprocess
        sys.exit(subprocess.call(launcher_command(args.gpus, sys.argv[1:])))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # CPU baseline FIRST (rank 0, N = 1 only), before the GPU is busy: the oracle's literal restatement of the reference prover (PCLMUL
    # gf192, one thread — the reference is single-threaded) on a bounded sample of the same workload; its transcript is compared with the
    # device prover's for that instance further down, before the number is reported.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.replay_rank is None:
        cpu = cpu_baseline_leg(args.cpu_log_n, args.cpu_all_cores, args.cpu_reps)

    import torch
    import torch.distributed as dist
    import libiop_amd
    from libiop_amd import aurora, domains, r1cs

    if args.gpus > 1 or world > 1 or args.force_sharded:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    lib = libiop_amd.lib()                      # raises when the HIP library is missing: no fallback
    lib.init(local_rank)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)

    field = domains.GF192()
    sharded = (world > 1 and not args.throughput) or args.force_sharded
    comm = None
    if sharded:
        from libiop_amd import dist as idist
        ops = idist.ShardedDeviceOps(lib, torch, dev, field, idist.AuroraShard(dist, rank, world))      # the Python prover's operator set: cross-check only
        comm = lib.comm_create_rccl_from_torch(dist, rank, world, dev)                                  # the native prover's RCCL communicator
    else:
        ops = domains.DeviceOps(lib, torch, dev, field)
    n = 1 << args.log_n
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, 15, n - 1, SEED)         # untimed: the statement and witness
    params = aurora.AuroraParameters(field, n, n - 1, 15)
    d_assignment = ops.upload(aurora.assignment_vector(field, primary, auxiliary))      # resident in HBM before the timed region
    torch.cuda.synchronize()

    # Every N: the native prover behind the C ABI (libiop_amd/cpp/aurora.hpp inside the library; iopx_aurora_prove on one GPU,
    # iopx_aurora_prove_dist over the communicator otherwise — the same code, libiop_amd/cpp/dist.hpp) on its own copy of the same seeded
    # instance; the Python prover (libiop_amd/aurora.py) proves it once below as a cross-check of the transcript bytes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    native = lib.aurora_example_instance(0, n, 15, n - 1, SEED)
    torch.cuda.synchronize()
    instance_create_s = time.perf_counter() - t0
    # one-time work ahead of the first proof (iopx_aurora_instance_warm: pool, plans, tables, transposed matrices), timed on its own; --no-warm leaves it
    # inside the first proof, as a caller that never calls it would see
    lib.cold_stats(reset=True)
    warm_s = None
    if not args.no_warm and comm is None and args.replay_rank is None:
        t0 = time.perf_counter()
        lib.aurora_instance_warm(native)
        torch.cuda.synchronize()
        warm_s = time.perf_counter() - t0
    one_time_costs = {k: {"count": v[0], "ms": round(v[1], 3)} for k, v in lib.cold_stats(reset=True).items()}

    if args.replay_rank is not None:
        # one rank of --world played alone: its line only (no headline value: the proof's bytes are meaningless)
        assert world == 1, "--replay-rank runs on one GPU"
        rr = rank_replay(lib, torch, native, params, args.world, args.replay_rank, steps=args.steps, warmup=args.warmup)
        print(json.dumps({"metric": "aurora_prover_rank_replay_ms", "value": rr["ms_per_proof"], "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "higher_is_better": False, "data": "synthetic", "config": {"workload": "rank %d of %d of the 2^%d Aurora prover over GF(2^192), played alone on one GPU "
                          "(compute path only: collectives completed locally, not timed)" % (args.replay_rank, args.world, args.log_n), "rank_replay": rr}}))
        lib.aurora_instance_free(native)
        return

    class _Bytes:
        def __init__(self, b):
            self.b = b

        def serialize(self):
            return self.b

    def step():
        if comm is not None:
            return _Bytes(lib.aurora_prove_dist(native, comm, 128, params.RS_extra_dimensions, 2))
        return _Bytes(lib.aurora_prove(native, 128, params.RS_extra_dimensions, 2))

    transcript = None
    import gc
    # the very first proof of this process on this instance, timed on its own: it builds what the later ones find cached — per-domain plans
    # and twiddle tables, the transposed lincheck matrices, subspace-polynomial tables, the buffer pool (reported, never part of `value`)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = step()
    torch.cuda.synchronize()
    first_proof_s = time.perf_counter() - t0
    first_proof_costs = {k: {"count": v[0], "ms": round(v[1], 3)} for k, v in lib.cold_stats(reset=True).items()}     # what the first proof still had to build
    for _ in range(args.warmup):
        transcript = step()
    if transcript is None:
        transcript = first
    gc.collect()
    gc.freeze()             # the instance, the plans and the modules are long-lived: keep them out of the collector's generations
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        transcript = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prover_s = dt / args.steps

    # the second, independently written prover (libiop_amd/aurora.py; over libiop_amd/dist.py's operators under --force-sharded) on the same
    # instance; with N > 1 ranks rank 0 proves the instance once more on its own GPU alone instead (no collective involved): the distributed
    # transcript must be the single-GPU prover's
    if world == 1:
        if not args.no_cross_check:
            check = aurora.aurora_snark_prover(ops, cs, primary, None, params, d_assignment=d_assignment).serialize()
            assert check == transcript.serialize(), "native prover's transcript differs from the Python prover's"
    elif rank == 0 and comm is not None:
        assert lib.aurora_prove(native, 128, params.RS_extra_dimensions, 2) == transcript.serialize(), "distributed transcript differs from the single-GPU prover's"

    # the same proof by the reference's own schedule (every virtual oracle over the whole codeword domain, coefficient forms: IOPX_HEAD_EVAL=0, read per
    # proof) — a second figure beside `value`, so that the reader sees what the schedule and what the kernels contribute; same transcript bytes
    reference_schedule = None
    if world == 1 and not args.no_secondary and lib.get_option("IOPX_HEAD_EVAL", 1) != 0:
        lib.set_option("IOPX_HEAD_EVAL", 0)             # looked up per proof
        try:
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                t_ref = step()
            torch.cuda.synchronize()
            ref_s = (time.perf_counter() - t0) / 5
        finally:
            lib.clear_option("IOPX_HEAD_EVAL")
        assert t_ref.serialize() == transcript.serialize(), "the two schedules produced different transcripts"
        reference_schedule = {"ms_per_step": ref_s * 1e3, "steps": 5, "transcript_equal": True}

    # per-kernel durations of one more proof, live, with HIP events on the stream the kernels are launched on.  The timed loop builds every round's
    # Merkle tree on the library's side stream, beside the next round's transforms; while the profiler records the library keeps those sections
    # on the main stream (iopx_side_stream_begin), so that every kernel is timed running alone and the durations add up.  The proof before it runs
    # the same way without the profiler (option IOPX_MERKLE_STREAM = 0, looked up per round): its wall time is what the kernel sum is compared with
    lib.comm_stats(reset=True)
    prev_merkle_stream = lib.get_option("IOPX_MERKLE_STREAM", 1)
    lib.set_option("IOPX_MERKLE_STREAM", 0)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        serial_proof_s = time.perf_counter() - t0
        lib.profile_begin()
        step()
        prof = lib.profile_report()
    finally:
        lib.set_option("IOPX_MERKLE_STREAM", prev_merkle_stream)
    comm_calls, comm_bytes = lib.comm_stats()
    dom_name, (dom_cnt, dom_ms, dom_bytes) = max(prof.items(), key=lambda kv: kv[1][1])
    dom_avg_s = dom_ms / dom_cnt / 1e3
    alg_bytes_per_launch = dom_bytes / dom_cnt if dom_bytes else None
    achieved = (dom_bytes / (dom_ms / 1e3) / 1e9) if dom_bytes else None
    # HBM bytes per launch of that kernel: NOT measured in this run (PMC counters need rocprofv3 passes of their own) — taken from the
    # committed PMC run of the same kernels and labelled with its file; null when that run covered other kernel sources
    traffic, traffic_source = None, None
    for tname in ("r06_traffic_aurora.json", "r05_traffic_aurora.json"):         # the newest collection whose sources are the ones that run now
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
            if tj.get("log_n") == args.log_n and dom_name in tj.get("kernels", {}) and tj.get("kernel_sources_sha256") == kernel_sources_digest():
                traffic = tj["kernels"][dom_name]["traffic_bytes_per_launch"]
                traffic_source = "profiles/%s (rocprofv3 --pmc passes of this command, collected %s)" % (tname, tj.get("collected", "?"))
                break
        except (OSError, ValueError):
            pass
    # ALU ceiling: the rate measured live in this run (field products of the launches / their HIP-event time) against the VALU-issue
    # ceiling of the kernel's instruction mix (tools/alu_model.py: ISA histogram x per-class cycles measured by tools/ubench/valu_rates)
    products = getattr(lib, "last_profile_products", {})
    device_products = sum(products.values())          # every field multiplication the profiled proof's kernels reported (this rank)
    model = alu_model()
    alu = {}
    if model:
        for kname, kmod in model["kernels"].items():
            # the model's kernels are templates; the profile has one name per instantiation that ran (k_bfly_upper_fwd / _inv, ...): the batch form
            # of the edge pass has an instruction mix of its own and is left out
            members = [k for k in prof if k.startswith(kname) and not k.endswith("_batch")]
            p_sum, ms_sum = sum(products.get(k, 0) for k in members), sum(prof[k][1] for k in members)
            if p_sum and ms_sum:
                rate = p_sum / (ms_sum / 1e3)
                alu[kname] = {"products_per_s": rate, "ceiling_products_per_s": kmod["alu_ceiling_products_per_s"],
                              "alu_ceiling_frac": rate / kmod["alu_ceiling_products_per_s"], "cycles_per_wave_product_model": kmod["cycles_per_wave_butterfly"]}
    fft_kernels = ("k_phase1", "k_bfly_upper", "k_bfly_edge", "k_pad_copy", "k_rs_combine", "k_fill")
    fft_ms = sum(v[1] for k, v in prof.items() if k.startswith(fft_kernels))

    final_dim = params.codeword_domain_dim - sum(params.localization_parameters)
    inventory = aurora_transform_inventory(args.log_n, params.RS_extra_dimensions, final_dim)
    mults = sum(ref_fft_ops(m)[0] for _, m in inventory)
    adds = sum(ref_fft_ops(m)[1] for _, m in inventory)
    value = (mults + adds) / prover_s
    if args.throughput and world > 1:
        value *= world                          # every rank finished one whole proof of its own in prover_s (max over the ranks)

    out = {
        "metric": "aurora_prover_fft_field_ops_per_s",
        "value": value,
        "unit": "field-ops/s",
        "unit_note": "numerator = the REFERENCE's operation count for the proof's transforms (eight zero-padded 2^25-point FFTs, ...: config.ref_fft_*), not "
                     "the device's own work: the device produces the same transcript bytes with far fewer products (config.device_field_products_per_proof) — the five "
                     "committed codewords and FRI's f_1 by cosets, every virtual oracle over the head of the codeword domain only (config.schedule)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": prover_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (args.throughput and world > 1) else "strong",
        "vs_baseline": None,
        "dtype": "gf2^192 (u32 VALU bit-ops)",
        "data": "synthetic",
        "config": {
            "workload": "Aurora SNARK prover, 2^%d-constraint synthetic R1CS over GF(2^192) (generate_r1cs_example(n, 15, n - 1), seed 0x%x), "
                        "security 128, RS_extra_dimensions 5, FRI localization 2, non-zk, BLAKE2b: one complete proof per step, "
                        "instance and witness resident in HBM" % (args.log_n, SEED),
            "log_n": args.log_n, "field": "gf192", "prover_s": prover_s,
            # this process's first timed proof of the instance: after iopx_aurora_instance_warm (instance_setup_ms) unless --no-warm, when it is the cold proof
            "first_proof_ms": first_proof_s * 1e3,
            "instance_setup_ms": (instance_create_s + (warm_s or 0.0)) * 1e3,
            "instance_setup": {"create_ms": instance_create_s * 1e3, "warm_ms": warm_s * 1e3 if warm_s is not None else None,
                               "what": "create: the seeded statement and witness built on the host and moved to HBM; warm: iopx_aurora_instance_warm — the device pool grown to "
                                       "a proof's footprint, plans / twiddle tables / per-domain tables / transposed lincheck matrices built (one discarded proof)",
                               "one_time_costs_ms": one_time_costs, "first_proof_one_time_costs_ms": first_proof_costs},
            "stages_ms": device_stages(prof),                # kernel time of one proof under the reference's profiling block names
            "prover": ("native: iopx_aurora_prove (libiop_amd/cpp/aurora.hpp behind the C ABI); transcript equal to libiop_amd/aurora.py's" if comm is None else
                       "native: iopx_aurora_prove_dist (libiop_amd/cpp/aurora.hpp + dist.hpp behind the C ABI, RCCL communicator of %d rank(s)); transcript equal to "
                       "the single-GPU prover's" % world),
            "schedule": ("reference's: every virtual oracle over the whole codeword domain (IOPX_HEAD_EVAL=0)" if lib.get_option("IOPX_HEAD_EVAL", 1) == 0 else
                         "virtual oracles over the head of the codeword domain (as many points as their polynomial has coefficients), f_1 folded there and re-extended, "
                         "confirmed on a second window; h, f_w and f_1 re-extended without coefficient forms (DESIGN.md section 4)"),
            "reference_schedule": reference_schedule,          # field_ops_per_s filled in below
            "device_field_products_per_proof": device_products if world == 1 else None,
            "device_field_products_per_proof_this_rank": device_products,
            "device_products_per_s": device_products * world / prover_s if device_products else None,
            "collectives_per_proof": comm_calls, "collective_payload_bytes_per_proof_this_rank": comm_bytes,
            "codeword_domain_dim": params.codeword_domain_dim, "localization": params.localization_parameters,
            "fri_query_repetitions": params.fri_query_repetitions, "pow_bits": params.pow_bits,
            "argument_bytes": len(transcript.serialize()) if transcript is not None else None,
            # tests/golden/oracle_aurora_transcript_digests_large.json holds the CPU oracle prover's digest for this instance (2^16, 2^18, 2^20)
            "transcript_blake2b": __import__("hashlib").blake2b(transcript.serialize(), digest_size=32).hexdigest() if transcript is not None else None,
            # ... and tests/golden/reference_over_shim.json what libiop's OWN prover produced for it (tests/harness: the reference's sources compiled unmodified over a
            # stand-in libff in the build container; 2^20: 4253 s on one core) — data, read here only to say whether the timed proof is that proof
            "transcript_equals_the_references_own_prover": reference_own_digest_equal(args.log_n, transcript),
            "ref_fft_mults_per_proof": mults, "ref_fft_adds_per_proof": adds,
            "fft_stage": {"ms": fft_ms, "field_ops_per_s": (mults + adds) / (fft_ms / 1e3) if fft_ms else None,
                          "note": "transform kernels only (k_phase1, k_bfly_upper, k_bfly_edge, padding): HIP-event time inside one proof"},
            "multi_gpu": ("throughput mode: %d independent proofs, one per rank, no collective (not BASELINE's sharded config: a labelled secondary figure)" % world
                          if (args.throughput and world > 1) else
                          "one proof sharded over %d ranks by contiguous cosets of every codeword (libiop_amd/cpp/dist.hpp)" % world if world > 1 else "single GPU"),
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                     "kernel": dom_name, "launches_per_step": dom_cnt, "avg_launch_ms": dom_avg_s * 1e3,
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "traffic_source": traffic_source,
                     "binding": "integer VALU issue (gfx950 has no carry-less multiply: a GF(2^192) product is ~450-1000 VALU ops)",
                     "alu_ceiling_frac": next((v["alu_ceiling_frac"] for k, v in alu.items() if dom_name.startswith(k)), None),
                     "alu": alu,
                     "note": "frac = algorithmic bytes / HBM peak as the contract asks; the kernel is bound by VALU issue, for which alu_ceiling_frac is the "
                             "figure: measured products/s over the ceiling of the kernel's instruction mix at the per-class issue costs measured on this GPU "
                             "(profiles/r06_alu_model.json, profiles/r05_valu_rates.txt)",
                     "kernels_ms_per_step": {k: round(v[1], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                     "kernel_launches_per_step": {k: v[0] for k, v in prof.items()},
                     "kernel_algorithmic_bytes_per_launch": {k: round(v[2] / v[0]) for k, v in prof.items() if v[2]},
                     # the profiled proof's kernel time (HIP events) against the timed loop's wall time per proof: launch gaps + host work + read-backs
                     "kernels_ms_total": round(sum(v[1] for v in prof.values()), 3),
                     # idle time between consecutive launches of the profiled proof (HIP events; the longest gaps name the kernels on both sides)
                     "launch_gaps": getattr(lib, "last_profile_gaps", None),
                     "host_gap_ms": round(prover_s * 1e3 - sum(v[1] for v in prof.values()), 3) if prof else None,
                     "host_gap_note": "timed loop's ms per proof minus the profiled proof's kernel sum; negative since round 5: the loop overlaps each round's Merkle tree with the "
                                      "next round's kernels on a second stream, the profiled proof (ms_per_proof_trees_on_main_stream) does not",
                     "ms_per_proof_trees_on_main_stream": round(serial_proof_s * 1e3, 3)},
    }

    if rank == 0 and world == 1 and not args.no_secondary and not args.no_rank_replay:
        # the N-rank prover's per-rank compute path, measured on this GPU: ranks 0 (which evaluates the heads) and N - 1 of 2, 4 and 8
        out["config"]["rank_replay"] = {
            "what": "iopx_aurora_prove_dist over iopx_comm_create_replay: one rank of N played alone on this GPU, collectives completed locally on the stream; "
                    "compute path only — no xGMI latency, no waiting for a slower peer; a proof's time over N GPUs is at least the max over its ranks",
            "single_gpu_ms": prover_s * 1e3,
            "ranks": [rank_replay(lib, torch, native, params, w, r) for w in (2, 4, 8) for r in (0, w - 1)]}

    if reference_schedule:
        reference_schedule["field_ops_per_s"] = (mults + adds) / (reference_schedule["ms_per_step"] / 1e3)
        reference_schedule["note"] = "IOPX_HEAD_EVAL=0: the reference's schedule on the same kernels (seven codeword extensions, virtual oracles over 2^25 points)"

    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE configs[1]: one additive FFT of 2^22 random coefficients on the standard-basis subspace, shift 0
        m = 22
        basis, shift = libiop_amd.standard_basis(m), np.zeros(3, dtype=np.uint64)
        d_in = ops.upload(np.random.Generator(np.random.PCG64(0x2201)).integers(0, 2**64, size=(1 << m, 3), dtype=np.uint64))
        d_out = ops.empty(1 << m)
        for _ in range(2):
            lib.additive_FFT_dev(d_in.data_ptr(), 1 << m, basis, shift, d_out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lib.additive_FFT_dev(d_in.data_ptr(), 1 << m, basis, shift, d_out.data_ptr())
        torch.cuda.synchronize()
        s = (time.perf_counter() - t0) / 10
        mu, ad = ref_fft_ops(m)
        out["config"]["secondary"] = {"workload": "configs[1]: additive FFT over GF(2^192), 2^22 coefficients, shift 0", "ms_per_step": s * 1e3,
                                      "field_ops_per_s": (mu + ad) / s}

    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE configs[2]: the FRI-only SNARK for a degree-2^20 polynomial on the 2^22-point codeword domain (RS_extra_dimensions 2, localization 2,
        # 1 interactive and 10 query repetitions: profiling/instrument_fri_snark.cpp:84-148), Merkle leaf hashing and proof of work included — the
        # native prover (iopx_fri_snark_prove: libiop_amd/cpp/fri.hpp inside the library), coefficients resident in HBM
        coeffs3 = ops.upload(r1cs.seeded_elements(field, 0x2203, 1 << 20))
        t3 = None
        for it in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr3 = lib.fri_snark_prove(0, coeffs3.data_ptr(), 1 << 20, 22, 2, 2, 1, 10)
            torch.cuda.synchronize()
            if it >= 2:
                t3 = min(t3, time.perf_counter() - t0) if t3 else time.perf_counter() - t0
        out["config"]["secondary_fri"] = {"workload": "configs[2]: FRI-only SNARK, degree 2^20 on the 2^22-point domain over GF(2^192), localization 2, incl. Merkle trees and "
                                                      "proof of work (native: iopx_fri_snark_prove)", "prover_ms_min": t3 * 1e3, "argument_bytes": len(tr3)}
        del coeffs3

    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE configs[4] on one GPU: the Fractal prover for a 2^20-constraint instance over the 181-bit field (k = 0 inputs, RS_extra 3,
        # localization 2: profiling/instrument_fractal_snark.cpp:93-120), the index (twelve 2^25-point oracles + their Merkle tree) built once
        # by the indexer and resident in HBM, as the reference's prover receives it (snark/fractal_snark.tcc:135-162)
        from libiop_amd import fractal
        f5 = domains.EdwardsFr()
        ops5 = domains.DeviceOps(lib, torch, dev, f5)
        cs5, prim5, aux5 = r1cs.generate_r1cs_example(ops5, n, 0, n - 1, 0x2205)
        params5 = fractal.FractalParameters(f5, cs5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index5, _ = fractal.fractal_snark_indexer(ops5, cs5, params5)
        torch.cuda.synchronize()
        indexer_s = time.perf_counter() - t0
        d_z5 = ops5.upload(aurora.assignment_vector(f5, prim5, aux5))
        tr5_py = fractal.fractal_snark_prover(ops5, index5, cs5, prim5, None, params5, d_assignment=d_z5).serialize()
        # the timed prover is the native one (iopx_fractal_index / iopx_fractal_prove: libiop_amd/cpp/fractal.hpp inside the library) on its
        # own copy of the same seeded instance; its transcript must equal the Python prover's
        inst5 = lib.aurora_example_instance(1, n, 0, n - 1, 0x2205)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.fractal_index(inst5)
        torch.cuda.synchronize()
        native_indexer_s = time.perf_counter() - t0
        warm5_s = None
        if not args.no_warm:
            t0 = time.perf_counter()
            lib.aurora_instance_warm(inst5, fractal=True)
            torch.cuda.synchronize()
            warm5_s = time.perf_counter() - t0
        times5 = []
        gc.collect()
        for it in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr5 = lib.fractal_prove(inst5)
            torch.cuda.synchronize()
            times5.append(time.perf_counter() - t0)
        assert tr5 == tr5_py, "native Fractal prover's transcript differs from the Python prover's"
        lib.profile_begin()                             # one more proof under the library's HIP-event profiler (side-stream sections stay on the main
        lib.fractal_prove(inst5)                        # stream while it records: every kernel timed alone), for the kernel breakdown
        prof5 = lib.profile_report()
        lib.aurora_instance_free(inst5)
        # configs[4] names 8 GPUs: ranks 0 and 7 of the residue-class distribution played alone on this GPU (compute path only, as config.rank_replay)
        fractal_replay = []
        if not args.no_rank_replay:
            for w, r in ((8, 0), (8, 7)):
                rinst = lib.aurora_example_instance(1, n, 0, n - 1, 0x2205)
                rcomm = lib.comm_create_replay(r, w)
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    lib.fractal_index_dist(rinst, rcomm)
                    torch.cuda.synchronize()
                    index_ms = (time.perf_counter() - t0) * 1e3
                    for _ in range(2):
                        lib.fractal_prove_dist(rinst, rcomm)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(4):
                        lib.fractal_prove_dist(rinst, rcomm)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / 4 * 1e3
                    lib.comm_stats(reset=True)
                    lib.profile_begin()
                    lib.fractal_prove_dist(rinst, rcomm)
                    rprof = lib.profile_report()
                    calls, payload = lib.comm_stats()
                    pow_ms = sum(v[1] for k, v in rprof.items() if k.startswith("k_pow_blake2b"))
                    fractal_replay.append({"world": w, "rank": r, "ms_per_proof": round(ms, 3), "ms_per_proof_pow_adjusted": round(ms - pow_ms * (w - 1) / w, 3),
                                           "indexer_ms_first_call": round(index_ms, 2), "kernels_ms_total": round(sum(v[1] for v in rprof.values()), 3),
                                           "kernels_ms": {k: round(v[1], 3) for k, v in sorted(rprof.items(), key=lambda kv: -kv[1][1])[:10]},
                                           "collectives_per_proof": calls, "collective_payload_bytes_this_rank": payload})
                finally:
                    lib.comm_destroy(rcomm)
                    lib.aurora_instance_free(rinst)
        out["config"]["secondary_fractal"] = {
            "workload": "configs[4] on 1 GPU: Fractal prover, 2^%d-constraint R1CS over the 181-bit field, k=0, codeword 2^%d" % (args.log_n, params5.codeword_domain_dim),
            "prover_ms": sorted(times5[1:])[len(times5[1:]) // 2] * 1e3, "prover_ms_min": min(times5[1:]) * 1e3, "prover_ms_all": [round(t * 1e3, 2) for t in times5], "indexer_ms_first_call": indexer_s * 1e3, "native_indexer_ms": native_indexer_s * 1e3,
            "warm_ms": warm5_s * 1e3 if warm5_s is not None else None, "first_proof_ms": times5[0] * 1e3,
            "prover": "native: iopx_fractal_prove (libiop_amd/cpp/fractal.hpp behind the C ABI); transcript equal to libiop_amd/fractal.py's",
            "argument_bytes": len(tr5), "fri_query_repetitions": params5.fri_query_repetitions,
            "kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof5.items(), key=lambda kv: -kv[1][1])[:10]},
            "rank_replay": fractal_replay}
        d5, (c5, ms5, b5) = max(prof5.items(), key=lambda kv: kv[1][1])
        traffic5, traffic5_source = None, None
        for tname in ("r06_traffic_fractal.json", "r05_traffic_fractal.json"):
            try:      # PMC passes of tools/fractal_bench.py at this size (tools/collect_profiles.sh), quoted only for the kernel sources they ran on
                tj5 = json.load(open(os.path.join(ROOT, "profiles", tname)))
                if tj5.get("log_n") == args.log_n and d5 in tj5.get("kernels", {}) and tj5.get("kernel_sources_sha256") == kernel_sources_digest():
                    traffic5 = tj5["kernels"][d5]["traffic_bytes_per_launch"]
                    traffic5_source = "profiles/%s (rocprofv3 --pmc passes of tools/fractal_bench.py --log-n %d, collected %s; averaged over the indexer's and the prover's launches)" % (tname, args.log_n, tj5.get("collected", "?"))
                    break
            except (OSError, ValueError):
                pass
        out["config"]["secondary_fractal"]["roofline"] = {
            "bound": "hbm", "kernel": d5, "launches_per_proof": c5, "avg_launch_ms": ms5 / c5, "algorithmic_bytes_per_launch": b5 / c5 if b5 else None,
            "achieved": (b5 / (ms5 / 1e3) / 1e9) if b5 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (b5 / (ms5 / 1e3) / 1e9 / HBM_PEAK_GBS) if b5 else None,
            "traffic": traffic5, "traffic_source": traffic5_source, "binding": "integer VALU issue (29-bit-limb Montgomery products in v_mad_u64_u32 accumulators)"}
        del index5, tr5, tr5_py, cs5, d_z5

    if cpu is not None:
        # the sample instance on the device: the transcript must equal the CPU oracle prover's byte for byte before its time is reported
        k = cpu["log_n"]
        nk = 1 << k
        cs_k, prim_k, aux_k = r1cs.generate_r1cs_example(ops, nk, 15, nk - 1, SEED)
        params_k = aurora.AuroraParameters(field, nk, nk - 1, 15)
        mine = aurora.aurora_snark_prover(ops, cs_k, prim_k, aux_k, params_k).serialize()
        assert mine == cpu["transcript"], "device transcript differs from the CPU oracle prover's"
        inv_k = aurora_transform_inventory(k, params_k.RS_extra_dimensions, params_k.codeword_domain_dim - sum(params_k.localization_parameters))
        ops_k = sum(sum(ref_fft_ops(m)) for _, m in inv_k)
        out["cpu_baseline"] = {"value": ops_k / cpu["seconds"], "unit": "field-ops/s", "cores": 1, "kind": "port",
                               "sample": "the same prover on a 2^%d-constraint instance (same protocol, rate 1/32, seed): median %.2f s of %d runs after one warm-up, on one "
                                         "core, run before the GPU loop; its transcript equals the device prover's byte for byte (headline_size: the 2^20 instance itself, "
                                         "measured once)" % (k, cpu["seconds"], len(cpu["runs_s"])),
                               "seconds": cpu["seconds"], "runs_s": cpu["runs_s"], "sample_log_n": k, "host": cpu["host"],
                               # inclusive seconds / calls of the sample under the reference's block names (nested: the FFT wrappers run inside other blocks)
                               "stages_s": {name: {"seconds": round(v[0], 6), "calls": v[1]} for name, v in sorted(cpu["blocks"].items())},
                               # the same prover at the headline size, measured once on this pool's host (not re-run here: 42 minutes of one core)
                               "headline_size": headline_cpu_figure()}
        if "all_cores" in cpu:
            ac = cpu["all_cores"]
            inv_a = aurora_transform_inventory(ac["log_n"], params_k.RS_extra_dimensions, ac["log_n"] + params_k.RS_extra_dimensions - sum(
                aurora.AuroraParameters(field, 1 << ac["log_n"], (1 << ac["log_n"]) - 1, 15).localization_parameters))
            out["cpu_baseline"]["all_cores_not_in_reference"] = {
                "value": ac["cores"] * sum(sum(ref_fft_ops(m)) for _, m in inv_a) / ac["seconds_for_one_proof_per_core"], "unit": "field-ops/s", "cores": ac["cores"],
                "sample": "one independent 2^%d proof in each of `cores` processes at once: %.2f s (the reference prover is single-threaded; this is an upper "
                          "bound on what a multi-threaded port could reach)" % (ac["log_n"], ac["seconds_for_one_proof_per_core"])}
    if rank == 0:
        print(json.dumps(out))
    if args.force_sharded and rank == 0 and world == 1:          # the distributed code path must produce the single-GPU prover's transcript
        assert lib.aurora_prove(native, 128, params.RS_extra_dimensions, 2) == transcript.serialize(), "sharded transcript differs from the single-GPU transcript"
        print("force-sharded: transcript equals the single-GPU prover's", file=sys.stderr)
    lib.aurora_instance_free(native)
    if comm is not None:
        lib.comm_destroy(comm)
    if world > 1 or args.force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
