"""Pins the oracle's binary-field arithmetic: irreducible moduli, PCLMUL == portable == big-int model,
field axioms.  (libff itself is absent from the reference tree; see oracle/field.hpp header.)"""
import numpy as np
import pytest

import oracle
from helpers import MODULI, clmul_int, from_int, gf_mul_int, polymod_int, rand_elems, to_int


def _polypowmod(base, e, mod):
    r = 1
    while e:
        if e & 1:
            r = polymod_int(clmul_int(r, base), mod)
        base = polymod_int(clmul_int(base, base), mod)
        e >>= 1
    return r


def _polygcd(a, b):
    while b:
        a, b = b, polymod_int(a, b)
    return a


@pytest.mark.parametrize("words", [1, 2, 3, 4])
def test_modulus_is_irreducible(words):
    # Rabin: x^(2^n) == x mod f, and gcd(x^(2^(n/p)) - x, f) == 1 for every prime p | n
    f, n = MODULI[words], 64 * words
    x = 2
    t = x
    frob = {}
    for i in range(1, n + 1):
        t = polymod_int(clmul_int(t, t), f)
        frob[i] = t
    assert frob[n] == x
    for p in {2, 3}:
        if n % p == 0:
            assert _polygcd(frob[n // p] ^ x, f) == 1


def test_clmul_pclmul_matches_portable():
    rng = np.random.default_rng(1)
    vals = [0, 1, 2**63, 2**64 - 1] + [int(v) for v in rng.integers(0, 2**64, size=200, dtype=np.uint64)]
    for a in vals[:40]:
        for b in vals[:40]:
            exp = clmul_int(a, b)
            assert oracle.clmul64(a, b) == (exp & (2**64 - 1), exp >> 64)
            assert oracle.clmul64(a, b, portable=True) == (exp & (2**64 - 1), exp >> 64)


@pytest.mark.parametrize("words", [1, 2, 3, 4])
def test_mul_matches_bigint_model(words):
    n = 300
    a, b = rand_elems(10 + words, n, words), rand_elems(20 + words, n, words)
    # edge words: all-ones, top bit, x^(n-1) * x
    a[0] = 0xFFFFFFFFFFFFFFFF
    b[0] = 0xFFFFFFFFFFFFFFFF
    a[1] = 0
    a[1, words - 1] = 1 << 63
    b[1] = 0
    b[1, 0] = 2
    out = oracle.gf_mul(a, b)
    for i in range(n):
        assert to_int(out[i]) == gf_mul_int(to_int(a[i]), to_int(b[i]), words)
    # x^(n-1) * x = tail of the modulus
    assert to_int(out[1]) == MODULI[words] ^ (1 << (64 * words))


@pytest.mark.parametrize("words", [1, 3])
def test_inverse_and_axioms(words):
    a, b, c = (rand_elems(s, 20, words) for s in (3, 4, 5))
    inv = oracle.gf_inv(a)
    one = np.zeros_like(a)
    one[:, 0] = 1
    assert np.array_equal(oracle.gf_mul(a, inv), one)
    # commutativity, associativity, distributivity
    assert np.array_equal(oracle.gf_mul(a, b), oracle.gf_mul(b, a))
    assert np.array_equal(oracle.gf_mul(oracle.gf_mul(a, b), c), oracle.gf_mul(a, oracle.gf_mul(b, c)))
    assert np.array_equal(oracle.gf_mul(a, b ^ c), oracle.gf_mul(a, b) ^ oracle.gf_mul(a, c))
