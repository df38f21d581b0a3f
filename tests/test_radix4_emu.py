"""The two-levels-per-trip form of the upper butterfly passes (IOPX_P2_RADIX4=1: four elements per lane in registers; off by default because it
measured slower on the MI355X) on the CPU-compiled kernels with the default tile geometry: the comb passes need 64-column tiles, which
transforms of 2^12 points and more have.  Forward, inverse and low-degree extension against the oracle."""
import os
import subprocess
import sys

SCRIPT = r"""
import numpy as np
import oracle
from emu_lib import emu
from helpers import rand_elems
W = 3
lib = emu()
for m in (12, 13, 15, 16):
    for kind in ("std", "general"):
        basis = oracle.standard_basis(m, W) if kind == "std" else rand_elems(50 + m, m, W)
        shift = np.array([1 << m, 0, 0], dtype=np.uint64) if kind == "std" else rand_elems(51 + m, 1, W)[0]
        coeffs = rand_elems(20 + m, 1 << m, W)
        ev = oracle.additive_fft(coeffs, basis, shift)
        assert np.array_equal(lib.additive_FFT(coeffs, basis, shift), ev), ("fft", m, kind)
        assert np.array_equal(lib.additive_IFFT(ev, basis, shift), coeffs), ("ifft", m, kind)
    short = rand_elems(30 + m, (1 << (m - 2)) - 5, W)
    assert np.array_equal(lib.additive_FFT(short, basis, shift), oracle.additive_fft(short, basis, shift)), ("lde", m)
print("ok")
"""


def test_two_levels_per_trip():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IOPX_P2_RADIX4="1", PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests")]))
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
