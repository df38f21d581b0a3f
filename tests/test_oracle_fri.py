"""The reference's FRI-aux tests on the oracle (libiop/tests/protocols/test_fri_aux.cpp:16-86, 126-173)."""
import numpy as np
import pytest

import oracle
from helpers import rand_elems


def _horner(coeffs, x):
    acc = np.zeros((1, coeffs.shape[1]), dtype=np.uint64)
    for i in range(coeffs.shape[0] - 1, -1, -1):
        acc = oracle.gf_mul(acc, x[None, :]) ^ coeffs[i][None, :]
    return acc[0]


@pytest.mark.parametrize("words", [1, 3])
@pytest.mark.parametrize("coset_size", [2, 4, 8])
def test_fold_of_low_degree_polynomial_is_constant(words, coset_size):
    # test_fri_aux.cpp:16-45: deg < coset size, dim-10 domain (reference: dim 15), standard basis + random
    # shift; every coset's interpolant evaluated at x equals P(x) (the reference asserts the first three).
    m = 10
    poly = rand_elems(words + coset_size, coset_size, words)
    basis = oracle.standard_basis(m, words)
    shift = rand_elems(77, 1, words)[0]
    evals = oracle.additive_fft(poly, basis, shift)
    x = rand_elems(78, 1, words)[0]
    nxt = oracle.fri_fold_additive(evals, basis, shift, coset_size, x)
    assert nxt.shape[0] == (1 << m) // coset_size
    assert (nxt == _horner(poly, x)[None, :]).all()


def test_fold_when_x_lies_in_the_domain():
    # fri_aux.tcc:77-86: x in coset -> the value of f at x
    words, m, cs = 3, 6, 4
    f = rand_elems(1, 1 << m, words)
    basis, shift = oracle.standard_basis(m, words), rand_elems(2, 1, words)[0]
    pts = oracle.all_subset_sums(basis, shift)
    nxt = oracle.fri_fold_additive(f, basis, shift, cs, pts[13])
    assert np.array_equal(nxt[13 // cs], f[13])


def test_query_position_expectations():
    # test_fri_aux.cpp:126-149 additive (gf64, dim 10, prev 2, cur 3)
    n, prev, cur = 1 << 10, 2, 3
    loc_n = n >> prev
    assert oracle.next_coset_query_positions(True, n, loc_n, 0, prev, cur) == [0, 1, 2, 3, 4, 5, 6, 7]
    assert oracle.next_coset_query_positions(True, n, loc_n, (1 << prev) * 5 + 1, prev, cur) == [0, 1, 2, 3, 4, 5, 6, 7]
    assert oracle.next_coset_query_positions(True, n, loc_n, (1 << prev) * 9 + 2, prev, cur) == list(range(8, 16))
    # test_fri_aux.cpp:151-173 multiplicative
    off = 1 << (10 - prev - cur)
    assert oracle.next_coset_query_positions(False, n, loc_n, 0, prev, cur) == [i * off for i in range(8)]
    assert oracle.next_coset_query_positions(False, n, loc_n, (1 << 9) + 5, prev, cur) == [5 + i * off for i in range(8)]


def test_localization_array():
    # fri_ldt.tcc:132-146; SURVEY.md §8 cfg3/cfg4
    assert oracle.localization_array(2, 22, 2) == [1] + [2] * 9
    assert oracle.localization_array(2, 25, 5) == [1] + [2] * 9


def test_domain_chain_is_consistent_with_fold():
    # fri_ldt.tcc:310-338: L^(i+1) = q(L^(i)); folding a codeword of a degree-<d polynomial gives a
    # codeword of degree < d/2^eta over L^(i+1)  (checked by IFFT over the derived domain).
    words, m, d = 3, 8, 32
    poly = rand_elems(11, d, words)
    basis, shift = oracle.standard_basis(m, words), np.zeros(words, dtype=np.uint64)
    loc = [1, 2]
    doms = oracle.fri_domains_additive(basis, shift, loc)
    cw, cur_b, cur_s = oracle.additive_fft(poly, basis, shift), basis, shift
    deg = d
    for i, eta in enumerate(loc):
        x = rand_elems(20 + i, 1, words)[0]
        cw = oracle.fri_fold_additive(cw, cur_b, cur_s, 1 << eta, x)
        cur_b, cur_s = doms[i]
        deg >>= eta
        coeffs = oracle.additive_ifft(cw, cur_b, cur_s)
        assert not coeffs[deg:].any()


@pytest.mark.parametrize("words", [1, 3])
def test_single_coset_verifier_formula_equals_the_whole_domain_fold(words):
    # test_fri_aux.cpp:47-86: evaluate_next_f_i_at_coset on one coset == the entry of evaluate_next_f_i_over_entire_domain,
    # including a point inside the coset (fri_aux.tcc:288-296)
    m, eta = 8, 3
    cs = 1 << eta
    f = rand_elems(5 + words, 1 << m, words)
    basis, shift = rand_elems(6, m, words), rand_elems(7, 1, words)[0]
    pts = oracle.all_subset_sums(basis, shift)
    for x in [rand_elems(8, 1, words)[0], pts[5 * cs + 3]]:
        whole = oracle.fri_fold_additive(f, basis, shift, cs, x)
        for j in [0, 5, (1 << m) // cs - 1]:
            got = oracle.fri_fold_at_coset(f[j * cs:(j + 1) * cs], basis[:eta], pts[j * cs], x)
            assert np.array_equal(got, whole[j]), (j,)
