"""Host-side cost of the provers: a 2^10 / 2^12 proof has next to no kernel time, so its wall time is orchestration + launch latency.
Python prover (libiop_amd/aurora.py), native prover (iopx_aurora_prove), and the Python prover on the sharded operator set with one rank."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libiop_amd
from libiop_amd import aurora, domains, r1cs

lib = libiop_amd.lib(); lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda:0")
field = domains.GF192()
ops = domains.DeviceOps(lib, torch, dev, field)
out = {}
for log_n in (10, 12, 14):
    n = 1 << log_n
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, 15, n - 1, 0x2204)
    params = aurora.AuroraParameters(field, n, n - 1, 15)
    d_z = ops.upload(aurora.assignment_vector(field, primary, auxiliary))
    inst = lib.aurora_example_instance(0, n, 15, n - 1, 0x2204)
    def timed(fn, reps=10):
        fn(); fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    py = timed(lambda: aurora.aurora_snark_prover(ops, cs, primary, None, params, d_assignment=d_z))
    nat = timed(lambda: lib.aurora_prove(inst))
    lib.profile_begin(); lib.aurora_prove(inst); prof = lib.profile_report()
    out[log_n] = {"python_ms": round(py, 2), "native_ms": round(nat, 2), "kernel_ms": round(sum(v[1] for v in prof.values()), 2), "launches": sum(v[0] for v in prof.values())}
    lib.aurora_instance_free(inst)
print(json.dumps(out))
