"""Where a process's FIRST proof spends its time beyond the kernels: one-time host-side costs by label (iopx_cold_stats), the kernel time of
the first proof against a warm one (HIP events), wall time of both.  Usage: python tools/cold_start.py [--fractal] [--log-n 20] [--warm]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, libiop_amd

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--fractal", action="store_true")
ap.add_argument("--warm", action="store_true", help="call iopx_aurora_instance_warm before the first proof")
a = ap.parse_args()
lib = libiop_amd.lib(); lib.init(0); lib.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << a.log_n
out = {"log_n": a.log_n, "prover": "fractal" if a.fractal else "aurora", "warm_call": a.warm}
lib.cold_stats(reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
inst = lib.aurora_example_instance(1, n, 0, n - 1, 0x2205) if a.fractal else lib.aurora_example_instance(0, n, 15, n - 1, 0x2204)
torch.cuda.synchronize(); out["instance_create_ms"] = (time.perf_counter() - t0) * 1e3
out["cold_in_create"] = lib.cold_stats(reset=True)
if a.fractal:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.fractal_index(inst)
    torch.cuda.synchronize(); out["index_ms"] = (time.perf_counter() - t0) * 1e3
    out["cold_in_index"] = lib.cold_stats(reset=True)
if a.warm and hasattr(lib, "aurora_instance_warm"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.aurora_instance_warm(inst, fractal=a.fractal)
    torch.cuda.synchronize(); out["warm_ms"] = (time.perf_counter() - t0) * 1e3
    out["cold_in_warm"] = lib.cold_stats(reset=True)
prove = (lambda: lib.fractal_prove(inst)) if a.fractal else (lambda: lib.aurora_prove(inst))
walls, kernel_sums = [], []
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.profile_begin(); prove(); prof = lib.profile_report()
    torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e3)
    kernel_sums.append(sum(v[1] for v in prof.values()))
    if i == 0:
        out["cold_in_first_proof"] = lib.cold_stats(reset=True)
        first_prof = prof
    last_prof = prof
out["proof_wall_ms_profiled"] = [round(w, 2) for w in walls]
out["proof_kernel_sum_ms"] = [round(k, 2) for k in kernel_sums]
out["cold_after_first"] = lib.cold_stats()
only_first = {k: (first_prof[k][0] - last_prof.get(k, (0, 0, 0))[0], round(first_prof[k][1] - last_prof.get(k, (0, 0, 0))[1], 3)) for k in first_prof
              if first_prof[k][0] != last_prof.get(k, (0, 0, 0))[0] or abs(first_prof[k][1] - last_prof.get(k, (0, 0, 0))[1]) > 0.2}
out["kernels_extra_in_first_proof (launches, ms)"] = only_first
print(json.dumps(out, indent=1))
