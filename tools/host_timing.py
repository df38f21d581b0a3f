import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, libiop_amd
lib = libiop_amd.lib(); lib.init(0); lib.set_stream(torch.cuda.current_stream().cuda_stream)
inst = lib.aurora_example_instance(0, 1 << 20, 15, (1 << 20) - 1, 0x2204)
for i in range(3):
    lib.aurora_prove(inst)
lib.set_option("IOPX_HOST_TIMING", 1)          # (the library asks the environment once per name: switch it through the option table)
torch.cuda.synchronize()
for i in range(2):
    t0 = time.perf_counter(); t = lib.aurora_prove(inst); t1 = time.perf_counter()
    print("call %.1f us" % ((t1 - t0) * 1e6), file=sys.stderr)
