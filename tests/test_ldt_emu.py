"""LDT-reducer kernels' logic on the CPU (product .hip sources compiled by tests/emu) against the oracle, plus the
reference's own identities on the oracle (tests/algebra/test_exponentiation.cpp:26-64, tests/protocols/test_ldt_reducer.cpp)."""
import numpy as np
import pytest

import ldt_cases as lc
import libiop_amd
import oracle
from emu_lib import emu
from helpers import rand_elems


@pytest.mark.parametrize("m,degrees,seed,kind", lc.ADDITIVE)
def test_additive(m, degrees, seed, kind):
    lc.check_additive(emu(), m, degrees, seed, kind)


@pytest.mark.parametrize("log_n,degrees,seed,shifted", lc.MULTIPLICATIVE)
def test_multiplicative(log_n, degrees, seed, shifted):
    lc.check_multiplicative(emu(), log_n, degrees, seed, shifted)


def test_errors():
    lc.check_errors(emu())


def test_oracle_subspace_element_powers_vs_naive():
    # test_exponentiation.cpp:26-44: linearised-polynomial powers equal libff::power per element
    basis, shift = rand_elems(5, 5, 3), rand_elems(6, 1, 3)[0]
    elems = oracle.all_subset_sums(basis, shift)
    for e in list(range(0, 40)) + [255, 256, 1000]:
        want = np.zeros_like(elems)
        want[:, 0] = 1
        sq, k = elems.copy(), e
        while k:
            if k & 1:
                want = oracle.gf_mul(want, sq)
            sq = oracle.gf_mul(sq, sq)
            k >>= 1
        assert np.array_equal(oracle.subspace_element_powers(basis, shift, e), want), e


def test_oracle_coset_element_powers_vs_naive():
    # test_exponentiation.cpp:46-64
    P = libiop_amd.EDWARDS_FR_MODULUS
    order = 32
    shift = libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR])[0]
    g = pow(libiop_amd.EDWARDS_FR_GENERATOR, (P - 1) // order, P)
    for e in [0, 1, 2, 3, 31, 32, 33, 1000]:
        want = [pow(libiop_amd.EDWARDS_FR_GENERATOR * pow(g, i, P) % P, e, P) for i in range(order)]
        assert np.array_equal(oracle.fp_coset_element_powers(order, shift, e), libiop_amd.edwards_to_montgomery(want))


def test_oracle_combination_is_the_stated_polynomial_identity():
    # evaluated_contents equals sum_k c_k f_k + sum_sub c' x^(shift) f_k evaluated pointwise (evaluation_at_point, :133-170)
    m, degrees = 4, [9, 4, 9, 2]
    basis, shift = rand_elems(7, m, 3), rand_elems(8, 1, 3)[0]
    evals = [rand_elems(20 + k, 1 << m, 3) for k in range(4)]
    r = rand_elems(9, 8, 3)
    got = oracle.ldt_combine_additive(evals, degrees, r, basis, shift)
    one = np.array([[1, 0, 0]], dtype=np.uint64)
    c = np.concatenate([one, r])
    want = np.zeros_like(evals[0])
    for k in range(4):
        want ^= oracle.gf_mul(np.repeat(c[k:k + 1], 1 << m, axis=0), evals[k])
    for i, k in enumerate([1, 3]):
        xp = oracle.subspace_element_powers(basis, shift, 9 - degrees[k])
        want ^= oracle.gf_mul(oracle.gf_mul(np.repeat(c[4 + i:5 + i], 1 << m, axis=0), xp), evals[k])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("m,h,seed,kind", [(5, 2, 1, "aurora"), (8, 3, 2, "general"), (7, 7, 3, "aurora"), (6, 0, 4, "general")])
def test_rowcheck_additive(m, h, seed, kind):
    lc.check_rowcheck_additive(emu(), m, h, seed, kind)


def test_rowcheck_is_a_polynomial_division():
    lc.check_rowcheck_is_a_polynomial_division(emu(), 8, 5, 3)


@pytest.mark.parametrize("log_n,log_h,seed", [(5, 2, 1), (8, 4, 2), (6, 6, 3), (7, 0, 4)])
def test_rowcheck_multiplicative(log_n, log_h, seed):
    lc.check_rowcheck_multiplicative(emu(), log_n, log_h, seed)


def test_rowcheck_errors():
    lc.check_rowcheck_errors(emu())


@pytest.mark.parametrize("m,idim,seed,kind", [(5, 2, 1, "aurora"), (9, 4, 2, "general"), (7, 0, 3, "general"), (10, 5, 4, "aurora")])
def test_fz_additive(m, idim, seed, kind):
    lc.check_fz_additive(emu(), m, idim, seed, kind)


@pytest.mark.parametrize("log_n,ilog,seed", [(5, 2, 1), (9, 4, 2), (13, 3, 3), (6, 0, 4)])
def test_fz_multiplicative(log_n, ilog, seed):
    lc.check_fz_multiplicative(emu(), log_n, ilog, seed)


@pytest.mark.parametrize("m,sdim,seed,kind", [(5, 2, 1, "aurora"), (9, 4, 2, "general"), (7, 3, 3, "unshifted"), (12, 6, 4, "aurora"), (3, 1, 5, "general")])
def test_sumcheck_g_additive(m, sdim, seed, kind):
    lc.check_sumcheck_g_additive(emu(), m, sdim, seed, kind)


@pytest.mark.parametrize("log_n,slog,seed", [(5, 2, 1), (9, 4, 2), (13, 3, 3), (6, 0, 4)])
def test_sumcheck_g_multiplicative(log_n, slog, seed):
    lc.check_sumcheck_g_multiplicative(emu(), log_n, slog, seed)


@pytest.mark.parametrize("n,k,seed,prime", [(16, 3, 1, False), (300, 1, 2, False), (1024, 3, 3, True), (7, 2, 4, True)])
def test_lincheck(n, k, seed, prime):
    lc.check_lincheck(emu(), n, k, seed, prime)


@pytest.mark.parametrize("m,degrees,seed,kind", lc.GAP1)
def test_ldt_combine_gap1_group(m, degrees, seed, kind):
    lc.check_additive(emu(), m, degrees, seed, kind)
