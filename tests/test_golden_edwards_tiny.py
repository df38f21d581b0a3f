"""The integer-only golden vectors of tests/golden/edwards_tiny.json (the 181-bit prime-field arm: multiplicative FFT / IFFT / known-degree IFFT, FRI
folds, a Merkle tree over multiplicative cosets) against (a) the oracle, (b) the product kernels compiled for the CPU, and (c) that the committed
JSON is what its generator writes.  The GPU leg is tests/test_gpu_parity.py::test_integer_only_golden_vectors_prime_field."""
import json
import os
import subprocess
import sys

import golden_cases_edwards as ge


def test_oracle_equals_the_integer_vectors():
    import oracle
    ge.check(lambda c, log_n, s: oracle.multiplicative_fft(c, 1 << log_n, s), oracle.multiplicative_ifft, oracle.multiplicative_ifft_known_degree,
             oracle.fri_fold_multiplicative, lambda o, cs: oracle.merkle_build(o, cs, False))


def test_cpu_compiled_kernels_equal_the_integer_vectors():
    import libiop_amd
    from emu_lib import emu
    lib = emu()
    ge.check(lib.multiplicative_FFT, lib.multiplicative_IFFT, lib.multiplicative_IFFT_of_known_degree, lib.multiplicative_evaluate_next_f_i,
             lambda o, cs: lib.merkle_tree(o, cs, domain_type=libiop_amd.DOMAIN_MULTIPLICATIVE))


def test_committed_vectors_are_reproducible(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    src = open(os.path.join(here, "golden", "make_edwards_tiny.py")).read().replace('os.path.dirname(os.path.abspath(__file__))', repr(str(tmp_path)))
    script = tmp_path / "gen.py"
    script.write_text(src)
    subprocess.check_call([sys.executable, str(script)])
    assert json.load(open(tmp_path / "edwards_tiny.json")) == ge.load()
