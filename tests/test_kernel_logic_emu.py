"""Kernel logic of the product sources, executed on the CPU (one thread per workgroup, tests/emu) and
compared bit-for-bit with the oracle.  The same comparisons run on the real GPU in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import oracle
import libiop_amd
from emu_lib import emu
from helpers import rand_elems, one_word_basis

W = 3


def test_field_mul():
    a, b = rand_elems(1, 500, W), rand_elems(2, 500, W)
    a[0] = 0xFFFFFFFFFFFFFFFF
    b[0] = 0xFFFFFFFFFFFFFFFF
    a[1] = 0
    assert np.array_equal(emu().gf192_mul(a, b), oracle.gf_mul(a, b))


def test_field_mul_uniform():
    a = rand_elems(3, 300, W)
    a[0] = 0xFFFFFFFFFFFFFFFF
    for c in (rand_elems(4, 1, W), np.full((1, W), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64), np.array([[1, 0, 0]], dtype=np.uint64),
              np.array([[0, 0, 1 << 63]], dtype=np.uint64)):
        assert np.array_equal(emu().gf192_mul(a, c), oracle.gf_mul(a, np.repeat(c, 300, axis=0)))


def _dom(m, kind, seed):
    if kind == "std0":
        return oracle.standard_basis(m, W), np.zeros(W, dtype=np.uint64)
    if kind == "aurora":      # standard basis, shift x^m (aurora_iop.tcc:282-287)
        return oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
    if kind == "stdrand":
        return oracle.standard_basis(m, W), rand_elems(seed, 1, W)[0]
    return rand_elems(seed + 1, m, W), rand_elems(seed, 1, W)[0]      # general basis (FRI-derived style)


@pytest.mark.parametrize("m", [1, 2, 3, 5, 8, 11, 12, 13])
@pytest.mark.parametrize("kind", ["std0", "aurora", "stdrand", "general"])
def test_fft_full_size(m, kind):
    if m >= 12 and kind in ("aurora", "stdrand"):
        pytest.skip("covered by std0/general at this size")
    basis, shift = _dom(m, kind, 100 + m)
    coeffs = rand_elems(m, 1 << m, W)
    assert np.array_equal(emu().additive_FFT(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift))


@pytest.mark.parametrize("m,ncoef", [(4, 0), (4, 1), (6, 2), (6, 5), (8, 16), (8, 100), (10, 255), (13, 300), (14, 4096), (15, 4097 * 2)])
def test_fft_lde(m, ncoef):
    basis, shift = _dom(m, "aurora", 7)
    coeffs = rand_elems(m + ncoef, ncoef, W)
    assert np.array_equal(emu().additive_FFT(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift))


@pytest.mark.parametrize("m,k,shift0,second", [(2, 0, 5, False), (3, 31, 0xFFFFFFFF, False), (7, 13, 0, False), (11, 1, 0x80000001, False),
                                               (12, 30, 77, False), (13, 7, 1 << 20, False), (3, 1, 3, True), (4, 31, 0xFFFFFFFF, True),
                                               (8, 17, 0x12345, True), (11, 2, 0xF0000000, True), (12, 30, 1, True)])
def test_one_word_last_level(m, k, shift0, second):
    basis = one_word_basis(m, k, 900 + m, second)
    shift = np.array([shift0, 0, 0], dtype=np.uint64)
    coeffs = rand_elems(70 + m, 1 << m, W)
    evals = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(emu().additive_FFT(coeffs, basis, shift), evals)
    assert np.array_equal(emu().additive_IFFT(evals, basis, shift), coeffs)
    # a low-degree extension: the first d vectors span the transform, the rest index the cosets (one-word too)
    if m >= 4:
        short = rand_elems(71 + m, 1 << (m - 2), W)
        assert np.array_equal(emu().additive_FFT(short, basis, shift), oracle.additive_fft(short, basis, shift))


@pytest.mark.parametrize("m,kind", [(17, "std0"), (18, "general")])
def test_phase1_relocated_twists(m, kind):
    """From 2^17 on a level's twist (levels 6..9) moves into the contiguous last pass of the level before it, where 64 consecutive elements
    share the multiplier (phase1_schedule): forward and inverse against the oracle."""
    basis, shift = _dom(m, kind, 300 + m)
    coeffs = rand_elems(m, 1 << m, W)
    evals = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(emu().additive_FFT(coeffs, basis, shift), evals)
    assert np.array_equal(emu().additive_IFFT(evals, basis, shift), coeffs)


@pytest.mark.parametrize("m", [1, 2, 4, 7, 11, 12, 13])
@pytest.mark.parametrize("kind", ["std0", "general"])
def test_ifft(m, kind):
    basis, shift = _dom(m, kind, 200 + m)
    evals = rand_elems(300 + m, 1 << m, W)
    assert np.array_equal(emu().additive_IFFT(evals, basis, shift), oracle.additive_ifft(evals, basis, shift))


def test_ifft_known_degree():
    m, deg = 9, 70
    basis, shift = _dom(m, "stdrand", 5)
    evals = oracle.additive_fft(rand_elems(1, deg, W), basis, shift)
    assert np.array_equal(emu().IFFT_of_known_degree(evals, deg, basis, shift), oracle.additive_ifft_known_degree(evals, deg, basis, shift))


@pytest.mark.parametrize("m,cs", [(1, 2), (3, 2), (6, 4), (8, 8), (10, 2), (10, 4), (7, 1)])
@pytest.mark.parametrize("kind", ["std0", "general"])
def test_fri_fold(m, cs, kind):
    basis, shift = _dom(m, kind, 400 + m)
    f = rand_elems(500 + m, 1 << m, W)
    x = rand_elems(600 + m, 1, W)[0]
    assert np.array_equal(emu().evaluate_next_f_i_over_entire_domain(f, basis, shift, cs, x),
                          oracle.fri_fold_additive(f, basis, shift, cs, x))


def test_fri_fold_x_in_domain():
    m, cs = 6, 4
    basis, shift = _dom(m, "general", 9)
    f = rand_elems(1, 1 << m, W)
    pts = oracle.all_subset_sums(basis, shift)
    for idx in (0, 13, 63):
        assert np.array_equal(emu().evaluate_next_f_i_over_entire_domain(f, basis, shift, cs, pts[idx]),
                              oracle.fri_fold_additive(f, basis, shift, cs, pts[idx]))


@pytest.mark.parametrize("additive", [True, False])
@pytest.mark.parametrize("r,cs,L", [(1, 1, 2), (1, 2, 16), (4, 2, 64), (1, 4, 32), (12, 2, 8), (2, 8, 4), (3, 2, 4096), (2, 2, 128), (2, 4, 64), (3, 4, 32), (4, 4, 256), (5, 2, 16)])
def test_merkle(additive, r, cs, L):
    n = L * cs
    oracles = [rand_elems(700 + k, n, W) for k in range(r)]
    got = emu().merkle_tree(oracles, cs, 0 if additive else 1)
    assert np.array_equal(got, oracle.merkle_build(oracles, cs, additive))


def test_merkle_zk_and_errors():
    oracles = [rand_elems(5, 64, W)]
    salts = np.random.default_rng(3).integers(0, 256, size=(32, 32), dtype=np.uint8)
    assert np.array_equal(emu().merkle_tree(oracles, 2, 0, salts), oracle.merkle_build(oracles, 2, True, salts))
    with pytest.raises(ValueError):             # merkle_tree.tcc:27-31 -> std::invalid_argument
        emu().merkle_tree([rand_elems(1, 2, W)], 2)
    with pytest.raises(ValueError):
        emu().merkle_tree([rand_elems(1, 12, W)], 2)
    with pytest.raises(ValueError):             # more coefficients than the domain holds
        emu().additive_FFT(rand_elems(1, 9, W), oracle.standard_basis(3, W), np.zeros(W, dtype=np.uint64))


# ---- multiplicative cosets over the 181-bit prime field --------------------------------------------------------
def _shifts():
    return [oracle.fp_one(), oracle.fp_from_ints([19])[0], oracle.fp_rand(5, 1)[0]]


def test_fp_generator_matches_oracle():
    import libiop_amd
    for k in (1, 5, 20, 31):
        assert np.array_equal(libiop_amd.edwards_subgroup_generator(k), oracle.fp_subgroup_generator(1 << k))


@pytest.mark.parametrize("logn", [1, 2, 3, 6, 9, 12, 13])
def test_mult_fft(logn):
    n = 1 << logn
    for ncoef in sorted({1, 2, 3, n // 2 + 1, n - 1, n}):
        if ncoef > n or ncoef < 1:
            continue
        coeffs = oracle.fp_rand(logn * 10 + ncoef, ncoef)
        for shift in _shifts()[: (3 if logn < 12 else 2)]:
            assert np.array_equal(emu().multiplicative_FFT(coeffs, logn, shift), oracle.multiplicative_fft(coeffs, n, shift)), (logn, ncoef)


def test_mult_fft_small_degree_large_domain():
    # 16 coefficients onto 2^14 points: only the top 4 index bits are active (Aurora's f_1v shape)
    coeffs = oracle.fp_rand(3, 16)
    shift = oracle.fp_from_ints([19])[0]
    assert np.array_equal(emu().multiplicative_FFT(coeffs, 14, shift), oracle.multiplicative_fft(coeffs, 1 << 14, shift))
    assert not emu().multiplicative_FFT(np.zeros((0, 3), dtype=np.uint64), 5, shift).any()


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 12, 13])
def test_mult_ifft(logn):
    n = 1 << logn
    ev = oracle.fp_rand(logn, n)
    for shift in _shifts()[:2]:
        assert np.array_equal(emu().multiplicative_IFFT(ev, shift), oracle.multiplicative_ifft(ev, shift))


def test_mult_ifft_known_degree():
    n, deg = 1 << 10, 100
    shift = oracle.fp_from_ints([19])[0]
    ev = oracle.multiplicative_fft(oracle.fp_rand(1, deg), n, shift)
    assert np.array_equal(emu().multiplicative_IFFT_of_known_degree(ev, deg, shift), oracle.multiplicative_ifft_known_degree(ev, deg, shift))


@pytest.mark.parametrize("logn,cs", [(1, 2), (4, 2), (6, 4), (8, 8), (10, 2), (10, 4), (5, 1)])
def test_mult_fri_fold(logn, cs):
    f = oracle.fp_rand(logn + cs, 1 << logn)
    x = oracle.fp_rand(99, 1)[0]
    for shift in _shifts()[:2]:
        assert np.array_equal(emu().multiplicative_evaluate_next_f_i(f, shift, cs, x), oracle.fri_fold_multiplicative(f, shift, cs, x))


def test_sharded_transform_building_blocks():
    import dist_blocks_check as c
    c.check_pow_table(emu())
    for log_n in (1, 2, 5, 10, 12):
        c.check_taylor(emu(), log_n)
    c.check_combine(emu())


@pytest.mark.parametrize("m,batch", [(1, 3), (5, 2), (9, 4), (12, 3)])
def test_batched_ifft(m, batch):
    lib = emu()
    basis, shift = rand_elems(40 + m, m, W), rand_elems(41 + m, 1, W)[0]
    ev = rand_elems(42 + m, batch << m, W)
    out = np.empty_like(ev)
    lib._check(lib.c.iopx_add_ifft_gf192_batch_dev(ev.ctypes.data, batch, basis.ctypes.data_as(libiop_amd._u64p), m,
                                                   shift.ctypes.data_as(libiop_amd._u64p), out.ctypes.data))
    for k in range(batch):
        assert np.array_equal(out[k << m:(k + 1) << m], oracle.additive_ifft(ev[k << m:(k + 1) << m], basis, shift)), k
    # in place
    lib._check(lib.c.iopx_add_ifft_gf192_batch_dev(ev.ctypes.data, batch, basis.ctypes.data_as(libiop_amd._u64p), m,
                                                   shift.ctypes.data_as(libiop_amd._u64p), ev.ctypes.data))
    assert np.array_equal(ev, out)


# the last five take the shared last pass (k_bfly_edge_fwd_batch: d >= 11, two to four polynomials; five = four + one)
@pytest.mark.parametrize("m,ncoef,batch,cb,cc", [(8, 50, 3, 0, 4), (10, 64, 2, 3, 5), (7, 128, 3, 0, 1), (6, 1, 2, 0, 64), (12, 1000, 4, 1, 2),
                                                 (13, 2048, 2, 1, 3), (13, 1500, 3, 0, 4), (12, 2048, 4, 0, 2), (14, 4096, 3, 2, 2), (13, 2048, 5, 0, 2)])
def test_batched_lde(m, ncoef, batch, cb, cc):
    lib = emu()
    basis, shift = rand_elems(50 + m, m, W), rand_elems(51 + m, 1, W)[0]
    d = 0 if ncoef <= 1 else int(np.ceil(np.log2(ncoef)))
    polys = [rand_elems(60 + k, ncoef, W) for k in range(batch)]
    outs = [np.zeros((cc << d, W), dtype=np.uint64) for _ in range(batch)]
    lib.additive_LDE_batch_dev([p.ctypes.data for p in polys], ncoef, basis, shift, cb, cc, [o.ctypes.data for o in outs])
    for k in range(batch):
        full = oracle.additive_fft(polys[k], basis, shift)
        assert np.array_equal(outs[k], full[cb << d:(cb + cc) << d]), k


@pytest.mark.parametrize("m,ncoef,batch,cb,cc", [(13, 2048, 3, 0, 4), (12, 2048, 2, 0, 2), (9, 128, 4, 0, 4), (13, 2048, 3, 1, 2), (14, 2048, 4, 5, 3)])
def test_batched_lde_standard_basis(m, ncoef, batch, cb, cc):
    """The prover's shape: standard basis, shift x^m — the shared last pass with one- and two-word twiddle numerators at pair bits 0 and 1,
    over the whole codeword and over a rank's coset range (the numerators of a coset's shift take its global index)."""
    lib = emu()
    basis, shift = _dom(m, "aurora", 0)
    d = int(np.ceil(np.log2(ncoef)))
    polys = [rand_elems(80 + k, ncoef, W) for k in range(batch)]
    outs = [np.zeros((cc << d, W), dtype=np.uint64) for _ in range(batch)]
    lib.additive_LDE_batch_dev([p.ctypes.data for p in polys], ncoef, basis, shift, cb, cc, [o.ctypes.data for o in outs])
    for k in range(batch):
        assert np.array_equal(outs[k], oracle.additive_fft(polys[k], basis, shift)[cb << d:(cb + cc) << d]), k


def test_half_wavefront_product_under_a_divergent_branch():
    """gf_mul_halves through its test entry on the CPU build: lanes 0..31 of a wavefront by the first multiplier, lanes 32..63 by the second, only
    where the branch around the product is taken (the EXEC part of the story is the GPU test's: tests/test_gpu_parity.py)."""
    import halves_cases
    halves_cases.check(emu(), None, count_active=False)


def test_noncanonical_prime_field_inputs_give_canonical_congruent_outputs():
    import noncanonical_cases
    noncanonical_cases.check(emu())
