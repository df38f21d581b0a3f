"""Degenerate sizes through the C ABI (CPU build of the kernels): single-element domains, empty inputs, whole-domain cosets —
the reference's asserts / exceptions for the same calls are cited."""
import numpy as np
import pytest

import libiop_amd as la
import oracle
from emu_lib import emu
from helpers import rand_elems

W = 3


def test_single_element_domain():
    lib = emu()
    b0, s = np.zeros((0, 3), dtype=np.uint64), rand_elems(1, 1, W)[0]
    c = rand_elems(2, 1, W)
    assert np.array_equal(lib.additive_FFT(c, b0, s), oracle.additive_fft(c, b0, s))
    assert np.array_equal(lib.additive_IFFT(c, b0, s), c)
    assert np.array_equal(lib.ldt_combine([c], [1], rand_elems(6, 2, W), b0, s), oracle.ldt_combine_additive([c], [1], rand_elems(6, 2, W), b0, s))
    sh = la.edwards_to_montgomery([19])[0]
    f = la.edwards_to_montgomery([5])
    assert np.array_equal(lib.multiplicative_FFT(f, 0, sh), oracle.multiplicative_fft(f, 1, sh))


def test_empty_polynomial_evaluates_to_zero():
    out = emu().additive_FFT(np.zeros((0, 3), dtype=np.uint64), oracle.standard_basis(3, W), rand_elems(1, 1, W)[0])
    assert out.shape == (8, 3) and not out.any()


def test_too_many_coefficients():
    with pytest.raises(ValueError):             # fft.tcc:48 asserts poly size <= domain size
        emu().additive_FFT(rand_elems(2, 9, W), oracle.standard_basis(3, W), rand_elems(1, 1, W)[0])


def test_fold_with_extreme_coset_sizes():
    lib = emu()
    basis, s = oracle.standard_basis(4, W), rand_elems(1, 1, W)[0]
    f, x = rand_elems(3, 16, W), rand_elems(4, 1, W)[0]
    for cs in (1, 16):                          # the identity-like fold and the whole domain as one coset
        assert np.array_equal(lib.evaluate_next_f_i_over_entire_domain(f, basis, s, cs, x), oracle.fri_fold_additive(f, basis, s, cs, x))
    with pytest.raises(ValueError):
        lib.evaluate_next_f_i_over_entire_domain(f, basis, s, 3, x)


def test_merkle_degenerate_shapes():
    lib = emu()
    f = rand_elems(3, 16, W)
    with pytest.raises(ValueError):             # a single leaf: merkle_tree.tcc:27-31
        lib.merkle_tree([f], 16)
    with pytest.raises(AssertionError):         # merkle_tree.tcc:98-108
        lib.merkle_tree([f], 0)


def test_proof_of_work_without_difficulty():
    assert emu().solve_pow(bytes(32), 0) == bytes(32)       # the challenge itself passes (pow.tcc:92-96)


# ---- the holographic prover's entry points: argument checks and degenerate sizes ----
def _ops(field_cls):
    import torch
    from libiop_amd import domains
    return domains.DeviceOps(emu(), torch, torch.device("cpu"), field_cls())


def test_div_argument_checks_and_empty_input():
    from libiop_amd import domains
    for cls in (domains.GF192, domains.EdwardsFr):
        ops = _ops(cls)
        a = ops.upload(rand_elems(3, 4, W) if cls is domains.GF192 else np.stack([cls().from_int(v) for v in (3, 5, 7, 11)]))
        with pytest.raises(ValueError):                     # the quotient buffer holds the running products: it cannot alias an input
            ops.lib.field_div_dev(a.data_ptr(), a.data_ptr(), a.data_ptr(), 4, prime_field=not cls().additive)
        ops.lib.field_div_dev(None, a.data_ptr(), a.data_ptr(), 0, prime_field=not cls().additive)          # nothing to do, nothing checked
        inv = ops.div(None, a)
        assert np.array_equal(ops.download(ops.mul(inv, a)), np.broadcast_to(cls().one(), (4, 3)))


def test_rational_combine_limits():
    from libiop_amd import domains
    ops = _ops(domains.GF192)
    v = [ops.upload(rand_elems(10 + i, 8, W)) for i in range(10)]
    with pytest.raises(ValueError):                         # at most four rationals (the shipped protocols combine three matrices)
        ops.rational_combine(v[:5], v[5:], rand_elems(1, 5, W), 8)
    with pytest.raises(ValueError):                         # one coefficient per rational
        ops.rational_combine(v[:3], v[3:6], rand_elems(1, 2, W), 8)
    N, D = ops.rational_combine(v[:1], v[1:2], rand_elems(2, 1, W), 8)                                     # one rational: N = c N_0, D = D_0
    assert np.array_equal(ops.download(D), ops.download(v[1]))


def test_sumcheck_constraint_needs_a_sub_domain():
    from libiop_amd import domains
    f = domains.GF192()
    ops = _ops(domains.GF192)
    L = f.domain(1 << 6, f.domain(1 << 6).element_outside_of_subset())
    other = domains.Domain(f, domains.ADDITIVE, basis=rand_elems(5, 3, W), shift=np.zeros(3, dtype=np.uint64))   # not a prefix of L's basis
    x = ops.upload(rand_elems(7, 64, W))
    with pytest.raises(ValueError):
        ops.rational_sumcheck_constraint(x, x, x, L, other, f.zero())
    fe = domains.EdwardsFr()
    opse = _ops(domains.EdwardsFr)
    small, big = fe.domain(1 << 4, 19), fe.domain(1 << 6)
    y = opse.upload(np.stack([fe.from_int(i + 1) for i in range(16)]))
    with pytest.raises(ValueError):                         # summation domain larger than the codeword domain
        opse.rational_sumcheck_constraint(y, y, y, small, big, fe.zero())


def test_codeword_domain_meeting_the_summation_domain_is_refused():
    """Z_K vanishes on the codeword domain when it is not shifted off K: the reference divides by zero there; refused here."""
    from libiop_amd import domains
    fe = domains.EdwardsFr()
    ops = _ops(domains.EdwardsFr)
    L, K = fe.domain(1 << 6), fe.domain(1 << 3)            # both unshifted: K is a subset of L
    y = ops.upload(np.stack([fe.from_int(i + 1) for i in range(64)]))
    with pytest.raises(ValueError):
        ops.rational_sumcheck_constraint(y, y, y, L, K, fe.zero())


def test_sums_of_products_at_the_limb_bounds():
    """The prime-field kernels add up to 8 products per Montgomery reduction in 64-bit column accumulators: the largest operands
    (p - 1 everywhere, and the all-ones limb pattern just below p) must not overflow them, for 8, 9 and 16 terms."""
    from libiop_amd import domains
    fe = domains.EdwardsFr()
    ops = _ops(domains.EdwardsFr)
    P = fe.P
    for value in (P - 1, P - 2, (1 << 180) - 1, P // 3):
        for terms in (1, 7, 8, 9, 16):
            vecs = [ops.upload(np.stack([fe.from_int(value)] * 5)) for _ in range(terms)]
            coeffs = np.stack([fe.from_int(value - i) for i in range(terms)])
            got = ops.download(ops.lincomb(vecs, coeffs, 5))
            expect = fe.from_int(sum(value * (value - i) for i in range(terms)))
            assert all(np.array_equal(row, expect) for row in got), (value, terms)
            got = ops.download(ops.lincomb_affine(vecs, coeffs, fe.from_int(P - 5), 5))
            expect = fe.from_int(sum(value * (value - i) for i in range(terms)) + P - 5)
            assert all(np.array_equal(row, expect) for row in got), (value, terms)
