"""cProfile of the host side of one Aurora 2^20 proof (after a warm-up proof): where the wall-clock not covered by kernels goes."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libiop_amd
from libiop_amd import aurora, domains, r1cs

lib = libiop_amd.lib(); lib.init(0)
dev = torch.device("cuda:0")
lib.set_stream(torch.cuda.current_stream().cuda_stream)
field = domains.GF192()
ops = domains.DeviceOps(lib, torch, dev, field)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, 15, n - 1, 0x2204)
params = aurora.AuroraParameters(field, n, n - 1, 15)
d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
for _ in range(2):
    aurora.aurora_snark_prover(ops, cs, primary, None, params, d_assignment=d_z)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
aurora.aurora_snark_prover(ops, cs, primary, None, params, d_assignment=d_z)
torch.cuda.synchronize()
pr.disable()
print("wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
