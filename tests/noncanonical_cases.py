"""Prime-field elements that are not canonical Montgomery representatives (raw 192-bit words, what libiop's random_vector<FieldT> produces — "invalid
elements for libff prime fields", algebra/polynomials/polynomial.tcc:233-234): outside the library's contract, but its behaviour on them is defined and
tested — the multiplicative transforms return CANONICAL words that are congruent mod p to the results a CPU path computes from the same raw words
(which carries unreduced representatives through its additions: two of libiop's Ligero tests depend on that and are not served)."""
import numpy as np

import libiop_amd
import oracle

P = libiop_amd.EDWARDS_FR_MODULUS


def _ints(a):
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) for r in a]


def check(lib, seed=5):
    rng = np.random.default_rng(seed)
    shift = libiop_amd.edwards_to_montgomery([3])[0]
    for m in (1, 4, 7):
        n = 1 << m
        raw = rng.integers(0, 2**64, size=(n, 3), dtype=np.uint64)                     # up to 2^192: about 2^11 p
        for got, want in ((lib.multiplicative_FFT(raw, m, shift), oracle.multiplicative_fft(raw, n, shift)),
                          (lib.multiplicative_IFFT(raw, shift), oracle.multiplicative_ifft(raw, shift))):
            g, w = _ints(got), _ints(want)
            assert all(x < P for x in g), "the kernels return canonical representatives"
            assert all((x - y) % P == 0 for x, y in zip(g, w)), "congruent mod p to the CPU path's result"
