"""LDT-reducer parity cases shared by the CPU-emulation suite and the GPU suite
(combined_LDT_virtual_oracle::evaluated_contents, ldt_reducer_aux.tcc:39-131)."""
import numpy as np
import pytest

import libiop_amd
import oracle
from helpers import rand_elems

W = 3


def _rand_fp(seed, n):
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(32), "little") % libiop_amd.EDWARDS_FR_MODULUS for _ in range(n)]
    return libiop_amd.edwards_to_montgomery(vals)


def check_additive(lib, m, degrees, seed, kind="random"):
    n = 1 << m
    if kind == "standard":
        basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
    else:
        basis, shift = rand_elems(seed + 1, m, W), rand_elems(seed + 2, 1, W)[0]
    evals = [rand_elems(seed + 10 + k, n, W) for k in range(len(degrees))]
    coeffs = rand_elems(seed + 3, 2 * len(degrees), W)
    got = lib.ldt_combine(evals, degrees, coeffs, basis, shift)
    assert np.array_equal(got, oracle.ldt_combine_additive(evals, degrees, coeffs, basis, shift))


def check_multiplicative(lib, log_n, degrees, seed, shifted=True):
    n = 1 << log_n
    gen = libiop_amd.edwards_subgroup_generator(log_n)
    shift = libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR if shifted else 1])[0]
    evals = [_rand_fp(seed + 10 + k, n) for k in range(len(degrees))]
    coeffs = _rand_fp(seed + 3, 2 * len(degrees))
    got = lib.ldt_combine_multiplicative(evals, degrees, coeffs, log_n, gen, shift)
    assert np.array_equal(got, oracle.ldt_combine_fp(evals, degrees, coeffs, n, shift))


def check_errors(lib):
    basis, shift = oracle.standard_basis(3, W), np.zeros(W, dtype=np.uint64)
    with pytest.raises(ValueError):     # ldt_reducer_aux.tcc:29-32
        lib.ldt_combine([rand_elems(1, 8, W)], [4], rand_elems(2, 3, W), basis, shift)
    with pytest.raises(ValueError):     # :56-59
        lib.ldt_combine([rand_elems(1, 8, W), rand_elems(1, 4, W)], [4, 4], rand_elems(2, 4, W), basis, shift)


ADDITIVE = [
    (1, [1], 0, "random"), (3, [5, 5, 5], 1, "random"), (6, [40, 17, 40, 33], 2, "random"), (9, [300, 44, 1, 299, 300], 3, "standard"),
    (10, [1 << 9, (1 << 9) - 1, 7], 4, "standard"), (4, [3, 9], 5, "random"),
    # Aurora-like spread: exponents 2^k - 1 (shared by three oracles), 2^k - 82 (shares 17 bits with it), 1, 2^k
    (8, [(1 << 7) - 1, 1 << 6, 1 << 6, 1 << 6, (1 << 6) + 21, (1 << 7) - 2, (1 << 6) - 1], 6, "standard"),
    # more distinct exponents than shared-power slots: the direct kernel
    (5, [200] + [200 - 3 * k - 1 for k in range(20)], 7, "random"),
    # all multi-bit exponents identical to the common set
    (6, [100, 100 - 7, 100 - 7, 100 - 15], 8, "random"),
]
MULTIPLICATIVE = [(1, [1], 0, True), (5, [20, 20], 1, True), (8, [100, 37, 100, 1], 2, True), (13, [5000, 4097, 3], 3, False)]


def _gf_pow(x, e):
    """x^e for a (count, 3) array, square-and-multiply through the oracle's product."""
    r = np.zeros_like(x)
    r[:, 0] = 1
    sq = x.copy()
    while e:
        if e & 1:
            r = oracle.gf_mul(r, sq)
        sq = oracle.gf_mul(sq, sq)
        e >>= 1
    return r


def check_additive_sampled(lib, m, degrees, seed, samples=256):
    """Full-size run checked pointwise at sampled positions against evaluation_at_point's formula (:133-170)."""
    n = 1 << m
    basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
    evals = [rand_elems(seed + 10 + k, n, W) for k in range(len(degrees))]
    coeffs = rand_elems(seed + 3, 2 * len(degrees), W)
    got = lib.ldt_combine(evals, degrees, coeffs, basis, shift)
    pos = np.random.default_rng(seed).integers(0, n, size=samples)
    pos[:4] = [0, 1, n - 1, n // 2]
    x = np.zeros((samples, W), dtype=np.uint64)
    x[:, 0] = pos.astype(np.uint64) ^ np.uint64(1 << m)          # standard basis: element j is the integer j, plus the shift
    one = np.array([[1, 0, 0]], dtype=np.uint64)
    c = np.concatenate([one, coeffs])
    rep = lambda v: np.repeat(v.reshape(1, W), samples, axis=0)
    want = np.zeros((samples, W), dtype=np.uint64)
    mx, sub = max(degrees), 0
    for k, d in enumerate(degrees):
        coef = rep(c[k])
        if d < mx:
            coef = coef ^ oracle.gf_mul(rep(c[len(degrees) + sub]), _gf_pow(x, mx - d))
            sub += 1
        want ^= oracle.gf_mul(coef, evals[k][pos])
    assert np.array_equal(got[pos], want)
