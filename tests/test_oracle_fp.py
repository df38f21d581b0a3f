"""Pins the oracle's prime-field layer (edwards_Fr, Montgomery words) and multiplicative-domain path:
constants re-derived with Python big ints, the reference's own tests (libiop/tests/algebra/test_fft.cpp:54-121:
multiplicative FFT == naive on subgroups and cosets, IFFT inverts; test_fri_aux.cpp:16-86 multiplicative arm)."""
import numpy as np
import pytest

import oracle

R = oracle.EDWARDS_R


def _is_probable_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_field_constants():
    assert R.bit_length() == 181 and _is_probable_prime(R)
    assert (R - 1) % (1 << 31) == 0 and (R - 1) % (1 << 32) != 0
    rou = pow(19, (R - 1) >> 31, R)
    assert rou == 695314865466598274460565335217615316274564719601897184
    assert pow(rou, 1 << 30, R) != 1 and pow(rou, 1 << 31, R) == 1            # order exactly 2^31
    assert oracle.fp_to_ints(oracle.fp_subgroup_generator(1 << 31)[None, :])[0] == rou
    assert oracle.fp_to_ints(oracle.fp_subgroup_generator(1 << 10)[None, :])[0] == pow(19, (R - 1) >> 10, R)


def test_montgomery_arithmetic_matches_bigints():
    rng = np.random.default_rng(5)
    xs = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(200)] + [0, 1, R - 1]
    ys = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(200)] + [R - 1, R - 1, R - 1]
    a, b = oracle.fp_from_ints(xs), oracle.fp_from_ints(ys)
    # Montgomery representative = x * 2^192 mod p, little-endian limbs
    for i in (0, 5, 201):
        assert int(a[i][0]) | (int(a[i][1]) << 64) | (int(a[i][2]) << 128) == xs[i] * (1 << 192) % R
    assert oracle.fp_to_ints(oracle.fp_mul(a, b)) == [x * y % R for x, y in zip(xs, ys)]
    assert oracle.fp_to_ints(oracle.fp_add(a, b)) == [(x + y) % R for x, y in zip(xs, ys)]
    assert oracle.fp_to_ints(oracle.fp_sub(a, b)) == [(x - y) % R for x, y in zip(xs, ys)]
    nz = a[:20]
    assert oracle.fp_to_ints(oracle.fp_mul(nz, oracle.fp_inv(nz))) == [1] * 20


@pytest.mark.parametrize("logn", range(1, 10))
def test_fft_equals_naive_incl_degree_aware_and_cosets(logn):
    n = 1 << logn
    one = oracle.fp_one()
    gen = oracle.fp_from_ints([19])[0]                        # coset shift = multiplicative generator (subgroup.tcc:311-315)
    rnd = oracle.fp_rand(77, 1)[0]
    for ncoef in sorted({1, 2, 3, n // 2, n - 1, n} - {0}):
        if ncoef > n:
            continue
        coeffs = oracle.fp_rand(logn * 100 + ncoef, ncoef)
        for shift in (one, gen, rnd):
            assert np.array_equal(oracle.multiplicative_fft(coeffs, n, shift), oracle.fp_naive_fft(coeffs, n, shift)), (logn, ncoef)


@pytest.mark.parametrize("logn", [1, 2, 5, 9])
def test_ifft_inverts_fft(logn):
    n = 1 << logn
    for shift in (oracle.fp_one(), oracle.fp_from_ints([19])[0]):
        coeffs = oracle.fp_rand(logn, n)
        ev = oracle.multiplicative_fft(coeffs, n, shift)
        assert np.array_equal(oracle.multiplicative_ifft(ev, shift), coeffs)


def test_ifft_of_known_degree():
    n, deg = 1 << 8, 40
    shift = oracle.fp_from_ints([19])[0]
    coeffs = oracle.fp_rand(3, deg)
    ev = oracle.multiplicative_fft(coeffs, n, shift)
    got = oracle.multiplicative_ifft_known_degree(ev, deg, shift)
    assert got.shape[0] == 64 and np.array_equal(got[:deg], coeffs) and not got[deg:].any()


def _horner(coeffs, x):
    acc = np.zeros((1, 3), dtype=np.uint64)
    for i in range(coeffs.shape[0] - 1, -1, -1):
        acc = oracle.fp_add(oracle.fp_mul(acc, x[None, :]), coeffs[i][None, :])
    return acc[0]


@pytest.mark.parametrize("cs", [2, 4, 8])
def test_fold_of_low_degree_polynomial_is_constant(cs):
    n = 1 << 9
    shift = oracle.fp_from_ints([19])[0]
    poly = oracle.fp_rand(cs, cs)
    ev = oracle.multiplicative_fft(poly, n, shift)
    x = oracle.fp_rand(9, 1)[0]
    nxt = oracle.fri_fold_multiplicative(ev, shift, cs, x)
    assert nxt.shape[0] == n // cs and (nxt == _horner(poly, x)[None, :]).all()
