"""Transforms with more cosets than the per-coset shift-term table holds (the prover's f_1v: 16 coefficients over 2^25 points): the twiddles take
their shift terms from per-byte tables of the coset index (k_rs_tables), the one- and two-word numerators from up to 24 coset bits.  The table's
cap is a process-wide tuning value, so the cases run in a child process with a small cap, on the CPU-compiled kernels."""
import os
import subprocess
import sys

SCRIPT = r"""
import numpy as np
import oracle
from emu_lib import emu
from helpers import rand_elems, one_word_basis
W = 3
lib = emu()
def check(basis, shift, ncoef, seed):
    coeffs = rand_elems(seed, ncoef, W)
    assert np.array_equal(lib.additive_FFT(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift)), (basis.shape, ncoef)
for m, ncoef in [(9, 2), (10, 3), (12, 16), (13, 5), (13, 200), (14, 16), (14, 3000)]:
    check(oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64), ncoef, 10 + m)       # one- and two-word numerators, many coset bits
    check(rand_elems(50 + m, m, W), rand_elems(51 + m, 1, W)[0], ncoef, 20 + m)                         # general basis: byte tables only
check(one_word_basis(13, 9, 7, True), np.array([0xDEADBEEF, 0, 0], dtype=np.uint64), 16, 3)
print("ok")
"""


def test_byte_tables_and_many_coset_bits():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IOPX_RS_COMB_CAP_LOG2="3", PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests")]))
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
