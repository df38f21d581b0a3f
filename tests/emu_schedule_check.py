"""Run by tests/test_kernel_schedules_emu.py in a subprocess with IOPX_TILE_BITS etc. set: small tiles make
the multi-pass phase-1 / phase-2 schedules (which a 2^22 transform uses with the default 4096-element
tiles) appear at sizes the CPU emulation finishes in seconds."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

import oracle  # noqa: E402
from emu_lib import emu  # noqa: E402
from helpers import rand_elems  # noqa: E402

W = 3
ms = [int(x) for x in sys.argv[1].split(",")]
for m in ms:
    basis = rand_elems(m, m, W)
    shift = rand_elems(50 + m, 1, W)[0]
    coeffs = rand_elems(100 + m, 1 << m, W)
    ev = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(emu().additive_FFT(coeffs, basis, shift), ev), ("fft", m)
    assert np.array_equal(emu().additive_IFFT(ev, basis, shift), coeffs), ("ifft", m)
    for d in {max(1, m - 5), max(1, m - 1)}:
        c2 = coeffs[: (1 << d) - 1]
        assert np.array_equal(emu().additive_FFT(c2, basis, shift), oracle.additive_fft(c2, basis, shift)), ("lde", m, d)
print("ok")
