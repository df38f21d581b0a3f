"""The reference's own FFT tests, run on the oracle (libiop/tests/algebra/test_fft.cpp:27-52, 123-139):
additive FFT == naive Horner evaluation at all_subset_sums order, IFFT inverts it — gf64 as the
reference tests it, plus gf192 (the field of BASELINE configs 2-4)."""
import numpy as np
import pytest

import oracle
from helpers import rand_elems


def _domain(seed, m, words, kind):
    if kind == "standard":
        basis = oracle.standard_basis(m, words)
    else:  # "random": independent basis vectors with overwhelming probability at these sizes
        basis = rand_elems(seed + 1000, m, words)
    shift = rand_elems(seed + 2000, 1, words)[0] if kind != "standard0" else np.zeros(words, dtype=np.uint64)
    return basis, shift


@pytest.mark.parametrize("words", [1, 3])
@pytest.mark.parametrize("m", list(range(1, 11)))
def test_additive_fft_equals_naive_and_ifft_inverts(words, m):
    # test_fft.cpp:27-52: random coefficients, standard basis + random shift
    coeffs = rand_elems(m * 7 + words, 1 << m, words)
    basis, shift = _domain(m, m, words, "standard")
    naive = oracle.naive_fft(coeffs, basis, shift)
    fft = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(naive, fft)
    assert np.array_equal(oracle.additive_ifft(naive, basis, shift), coeffs)


@pytest.mark.parametrize("words", [1, 3])
def test_general_basis_and_fewer_coefficients(words):
    for m in (3, 6, 8):
        basis, shift = _domain(50 + m, m, words, "random")
        for ncoef in (1, 3, (1 << m) // 2, (1 << m) - 1):
            coeffs = rand_elems(m + ncoef, ncoef, words)
            assert np.array_equal(oracle.naive_fft(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift))


def test_roundtrip_large_gf64():
    # test_fft.cpp:123-139 (round trip up to 2^21 in the reference; 2^16 keeps the CPU suite short)
    for m in (12, 16):
        coeffs = rand_elems(m, 1 << m, 1)
        basis, shift = _domain(m, m, 1, "standard")
        assert np.array_equal(oracle.additive_ifft(oracle.additive_fft(coeffs, basis, shift), basis, shift), coeffs)


def test_lde_block_structure_gf192():
    # SURVEY.md §7 "LDE structure": block t (size 2^d) of a zero-padded transform over 2^m points is the
    # size-2^d FFT over span(basis[0..d)) with shift element_by_index(t * 2^d).
    words, d, m = 3, 5, 8
    coeffs = rand_elems(99, 1 << d, words)
    basis = oracle.standard_basis(m, words)
    shift = np.array([1 << m, 0, 0], dtype=np.uint64)           # Aurora-style shift x^m
    full = oracle.additive_fft(coeffs, basis, shift)
    pts = oracle.all_subset_sums(basis, shift)
    for t in range(1 << (m - d)):
        blk = oracle.additive_fft(coeffs, basis[:d], pts[t << d])
        assert np.array_equal(full[t << d:(t + 1) << d], blk)


def test_ifft_of_known_degree():
    words, m, deg = 3, 8, 40        # -> first 64 evaluations (fft.tcc:458-475)
    coeffs = rand_elems(5, deg, words)
    basis = oracle.standard_basis(m, words)
    shift = rand_elems(6, 1, words)[0]
    evals = oracle.additive_fft(coeffs, basis, shift)
    got = oracle.additive_ifft_known_degree(evals, deg, basis, shift)
    assert got.shape[0] == 64
    assert np.array_equal(got[:deg], coeffs) and not got[deg:].any()
