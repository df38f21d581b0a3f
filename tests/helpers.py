"""Shared helpers for the test-suite: seeded inputs and an independent big-int GF(2^n) model."""
import numpy as np

# modulus tails restated from libff's published binary fields (see oracle/field.hpp header)
MODULI = {
    1: (1 << 64) | 0x1B,
    2: (1 << 128) | 0x87,
    3: (1 << 192) | 0x87,
    4: (1 << 256) | 0x425,
}


def splitmix64(seed, count):
    """SplitMix64 stream (seeded synthetic inputs), returned as uint64 array."""
    out = np.empty(count, dtype=np.uint64)
    x = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(count):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        out[i] = z ^ (z >> 31)
    return out


def rand_elems(seed, count, words):
    """count uniformly random field elements as (count, words) uint64 (fast, numpy Philox-free)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 2**64, size=(count, words), dtype=np.uint64)


def to_int(e):
    v = 0
    for i, w in enumerate(e):
        v |= int(w) << (64 * i)
    return v


def from_int(v, words):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(words)], dtype=np.uint64)


def clmul_int(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return r


def polymod_int(a, mod):
    dm = mod.bit_length() - 1
    while a.bit_length() - 1 >= dm and a:
        a ^= mod << (a.bit_length() - 1 - dm)
    return a


def gf_mul_int(a, b, words):
    return polymod_int(clmul_int(a, b), MODULI[words])


def one_word_basis(m, k, seed, second=False):
    """m independent one-word vectors, the last one x^k: the shape that takes the one-word last level (gf_mul_small_over_xk); with
    `second` the one before it is x^(k-1), which puts the level above on two-word numerators too (gf_mul_small2_over)."""
    rng = np.random.default_rng(seed)
    while True:
        vals = [int(v) for v in rng.integers(1, 1 << 32, size=m - 1)] + [1 << k]
        if second:
            vals[m - 2] = 1 << (k - 1)
        rows, rank = list(vals), 0
        for bit in range(32):
            piv = next((i for i in range(rank, m) if (rows[i] >> bit) & 1), None)
            if piv is None:
                continue
            rows[rank], rows[piv] = rows[piv], rows[rank]
            rows = [r ^ rows[rank] if i != rank and (r >> bit) & 1 else r for i, r in enumerate(rows)]
            rank += 1
        if rank == m:
            basis = np.zeros((m, 3), dtype=np.uint64)
            basis[:, 0] = vals
            return basis
