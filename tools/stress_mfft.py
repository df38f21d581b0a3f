"""Stress of the host-pointer multiplicative transforms against the oracle: thousands of small calls with changing sizes, lengths and shifts in one process
(the call pattern of libiop's Ligero tests: tests/harness reftests on the GPU showed one flaky case).  Prints the first mismatches with their parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libiop_amd, oracle

lib = libiop_amd.lib(); lib.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
P = libiop_amd.EDWARDS_FR_MODULUS
def elems(n):
    return libiop_amd.edwards_to_montgomery([int.from_bytes(rng.bytes(24), "little") % P for _ in range(n)])
shifts = [libiop_amd.edwards_to_montgomery([1])[0], libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR])[0]] + [elems(1)[0] for _ in range(6)]
bad = 0
for it in range(iters):
    m = int(rng.integers(1, 9))
    n = 1 << m
    shift = shifts[int(rng.integers(0, len(shifts)))]
    kind = int(rng.integers(0, 3))
    if kind == 0:
        nc = int(rng.integers(1, n + 1))
        c = elems(nc)
        got, want = lib.multiplicative_FFT(c, m, shift), oracle.multiplicative_fft(c, n, shift)
    elif kind == 1:
        v = elems(n)
        got, want = lib.multiplicative_IFFT(v, shift), oracle.multiplicative_ifft(v, shift)
    else:
        if m < 2: continue
        v = elems(n); x = elems(1)[0]; cs = 2 if m < 3 else int(rng.choice([2, 4]))
        got, want = lib.multiplicative_evaluate_next_f_i(v, shift, cs, x), oracle.fri_fold_multiplicative(v, shift, cs, x)
    if not np.array_equal(got, want):
        bad += 1
        if bad <= 10:
            print("MISMATCH it", it, "kind", kind, "m", m, "first diff row", int(np.argmax((got != want).any(axis=1))), "rows differing", int((got != want).any(axis=1).sum()), flush=True)
print("iterations", iters, "mismatches", bad)
