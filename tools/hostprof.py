"""Host-side (Python / ctypes) profile of one proof on the GPU: where the wall-clock time that is not kernel time goes."""
import argparse, cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libiop_amd
from libiop_amd import aurora, domains, fractal, r1cs

ap = argparse.ArgumentParser()
ap.add_argument("--protocol", default="fractal")
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()
lib = libiop_amd.lib(); lib.init(0); dev = torch.device("cuda:0"); lib.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << a.log_n
if a.protocol == "aurora":
    f = domains.GF192(); ops = domains.DeviceOps(lib, torch, dev, f)
    cs, pr, aux = r1cs.generate_r1cs_example(ops, n, 15, n - 1, 0x2204)
    params = aurora.AuroraParameters(f, n, n - 1, 15)
    dz = ops.upload(aurora.assignment_vector(f, pr, aux))
    run = lambda: aurora.aurora_snark_prover(ops, cs, pr, None, params, d_assignment=dz)
else:
    f = domains.EdwardsFr(); ops = domains.DeviceOps(lib, torch, dev, f)
    cs, pr, aux = r1cs.generate_r1cs_example(ops, n, 0, n - 1, 0x2205)
    params = fractal.FractalParameters(f, cs)
    index, _ = fractal.fractal_snark_indexer(ops, cs, params)
    dz = ops.upload(aurora.assignment_vector(f, pr, aux))
    run = lambda: fractal.fractal_snark_prover(ops, index, cs, pr, None, params, d_assignment=dz)
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("wall ms", (time.perf_counter() - t0) * 1e3)
p = cProfile.Profile(); p.enable(); run(); torch.cuda.synchronize(); p.disable()
pstats.Stats(p).sort_stats("tottime").print_stats(a.top)
