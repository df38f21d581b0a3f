"""Parity tests proper: the HIP path, called through the C ABI on a real MI355X, against the CPU oracle
(bit-exact) on seeded inputs, plus size-independent properties at BASELINE.json's full sizes."""
import numpy as np
import pytest

import oracle
from helpers import rand_elems, one_word_basis

pytestmark = pytest.mark.gpu
W = 3


@pytest.fixture(scope="module")
def gpu():
    import libiop_amd
    lib = libiop_amd.lib()          # raises if the HIP library is missing: no fallback
    lib.init(0)
    return lib


def _dom(m, kind, seed):
    if kind == "std0":
        return oracle.standard_basis(m, W), np.zeros(W, dtype=np.uint64)
    if kind == "aurora":
        return oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
    return rand_elems(seed + 1, m, W), rand_elems(seed, 1, W)[0]


def test_integer_only_golden_vectors(gpu):
    """tests/golden/gf192_tiny.json: FFT / IFFT / LDE / folds / a Merkle tree computed with Python integers and hashlib only."""
    import golden_cases
    golden_cases.check(gpu.additive_FFT, gpu.additive_IFFT, gpu.evaluate_next_f_i_over_entire_domain, gpu.merkle_tree)


def test_integer_only_golden_vectors_prime_field(gpu):
    """tests/golden/edwards_tiny.json: multiplicative FFT / IFFT / known-degree IFFT / folds / a Merkle tree over multiplicative cosets for the 181-bit
    field, computed with Python integers and hashlib only (tests/golden/make_edwards_tiny.py) — the HIP path against vectors that share no code with the oracle."""
    import golden_cases_edwards
    import libiop_amd
    golden_cases_edwards.check(gpu.multiplicative_FFT, gpu.multiplicative_IFFT, gpu.multiplicative_IFFT_of_known_degree, gpu.multiplicative_evaluate_next_f_i,
                               lambda o, cs: gpu.merkle_tree(o, cs, domain_type=libiop_amd.DOMAIN_MULTIPLICATIVE))


def test_field_mul(gpu):
    a, b = rand_elems(1, 1 << 16, W), rand_elems(2, 1 << 16, W)
    a[0] = 0xFFFFFFFFFFFFFFFF
    b[0] = 0xFFFFFFFFFFFFFFFF
    a[1] = 0
    a[2] = 0
    a[2, 2] = 1 << 63
    b[2] = 0
    b[2, 0] = 2
    assert np.array_equal(gpu.gf192_mul(a, b), oracle.gf_mul(a, b))


def test_field_mul_uniform(gpu):
    a = rand_elems(3, 1 << 14, W)
    a[0] = 0xFFFFFFFFFFFFFFFF
    for c in (rand_elems(4, 1, W), np.full((1, W), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64), np.array([[1, 0, 0]], dtype=np.uint64),
              np.array([[0, 0, 1 << 63]], dtype=np.uint64)):
        assert np.array_equal(gpu.gf192_mul(a, c), oracle.gf_mul(a, np.repeat(c, a.shape[0], axis=0)))


@pytest.mark.parametrize("m", [1, 2, 3, 6, 10, 11, 12, 13, 16, 18])
@pytest.mark.parametrize("kind", ["std0", "aurora", "general"])
def test_fft_full_size(gpu, m, kind):
    basis, shift = _dom(m, kind, 100 + m)
    coeffs = rand_elems(m, 1 << m, W)
    assert np.array_equal(gpu.additive_FFT(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift))


@pytest.mark.parametrize("m,k,shift0,second", [(2, 0, 5, False), (3, 31, 0xFFFFFFFF, False), (11, 1, 0x80000001, False), (13, 7, 1 << 20, False),
                                               (16, 20, 0xABCDEF01, False), (3, 1, 3, True), (4, 31, 0xFFFFFFFF, True), (12, 30, 1, True), (16, 18, 0xFFFF0000, True)])
def test_one_word_last_levels(gpu, m, k, shift0, second):
    """One-word bases ending in x^k (and x^(k-1), x^k): the last one / two butterfly levels multiply by one- / two-word numerators and
    divide exactly (gf_mul_small_over_xk, gf_mul_small2_over) — forward, inverse and as a low-degree extension over one-word cosets."""
    basis = one_word_basis(m, k, 900 + m, second)
    shift = np.array([shift0, 0, 0], dtype=np.uint64)
    coeffs = rand_elems(70 + m, 1 << m, W)
    evals = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(gpu.additive_FFT(coeffs, basis, shift), evals)
    assert np.array_equal(gpu.additive_IFFT(evals, basis, shift), coeffs)
    if m >= 4:
        short = rand_elems(71 + m, 1 << (m - 2), W)
        assert np.array_equal(gpu.additive_FFT(short, basis, shift), oracle.additive_fft(short, basis, shift))


@pytest.mark.parametrize("m,ncoef", [(4, 0), (4, 1), (6, 5), (10, 255), (13, 300), (15, 4096), (17, 8195), (18, 1 << 13), (20, (1 << 15) - 1)])
def test_fft_lde(gpu, m, ncoef):
    basis, shift = _dom(m, "aurora", 7)
    coeffs = rand_elems(m + ncoef, ncoef, W)
    assert np.array_equal(gpu.additive_FFT(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift))


@pytest.mark.parametrize("m", [1, 2, 5, 11, 12, 13, 16, 18])
@pytest.mark.parametrize("kind", ["std0", "general"])
def test_ifft(gpu, m, kind):
    basis, shift = _dom(m, kind, 200 + m)
    evals = rand_elems(300 + m, 1 << m, W)
    assert np.array_equal(gpu.additive_IFFT(evals, basis, shift), oracle.additive_ifft(evals, basis, shift))


def test_ifft_known_degree(gpu):
    m, deg = 14, 3000
    basis, shift = _dom(m, "aurora", 5)
    evals = oracle.additive_fft(rand_elems(1, deg, W), basis, shift)
    assert np.array_equal(gpu.IFFT_of_known_degree(evals, deg, basis, shift), oracle.additive_ifft_known_degree(evals, deg, basis, shift))


@pytest.mark.parametrize("m,cs", [(1, 2), (3, 2), (6, 4), (8, 8), (14, 2), (14, 4), (7, 1), (16, 4)])
@pytest.mark.parametrize("kind", ["std0", "general"])
def test_fri_fold(gpu, m, cs, kind):
    basis, shift = _dom(m, kind, 400 + m)
    f = rand_elems(500 + m, 1 << m, W)
    x = rand_elems(600 + m, 1, W)[0]
    assert np.array_equal(gpu.evaluate_next_f_i_over_entire_domain(f, basis, shift, cs, x),
                          oracle.fri_fold_additive(f, basis, shift, cs, x))


def test_fri_fold_x_in_domain(gpu):
    m, cs = 6, 4
    basis, shift = _dom(m, "general", 9)
    f = rand_elems(1, 1 << m, W)
    pts = oracle.all_subset_sums(basis, shift)
    for idx in (0, 13, 63):
        assert np.array_equal(gpu.evaluate_next_f_i_over_entire_domain(f, basis, shift, cs, pts[idx]),
                              oracle.fri_fold_additive(f, basis, shift, cs, pts[idx]))


def test_fri_chain_cfg3_shape_small(gpu):
    # the cfg3 round structure ([1,2,2,...], fri_ldt.tcc:132-146) on a dim-12 codeword, derived domains included
    m, rs = 12, 2
    loc = oracle.localization_array(2, m, rs)
    basis, shift = _dom(m, "std0", 0)
    doms = oracle.fri_domains_additive(basis, shift, loc)
    cw_g = cw_o = oracle.additive_fft(rand_elems(3, 1 << (m - rs), W), basis, shift)
    cb, cs_ = basis, shift
    for i, eta in enumerate(loc):
        x = rand_elems(900 + i, 1, W)[0]
        cw_g = gpu.evaluate_next_f_i_over_entire_domain(cw_g, cb, cs_, 1 << eta, x)
        cw_o = oracle.fri_fold_additive(cw_o, cb, cs_, 1 << eta, x)
        assert np.array_equal(cw_g, cw_o)
        cb, cs_ = doms[i]


@pytest.mark.parametrize("additive", [True, False])
@pytest.mark.parametrize("r,cs,L", [(1, 1, 2), (1, 2, 16), (4, 2, 64), (1, 4, 32), (12, 2, 8), (2, 8, 4), (4, 2, 1 << 15), (1, 2, 1 << 17), (2, 2, 128), (2, 4, 1 << 12), (3, 4, 32), (4, 4, 256), (3, 2, 1 << 14), (5, 2, 16)])
def test_merkle(gpu, additive, r, cs, L):
    n = L * cs
    oracles = [rand_elems(700 + k, n, W) for k in range(r)]
    got = gpu.merkle_tree(oracles, cs, 0 if additive else 1)
    assert np.array_equal(got, oracle.merkle_build(oracles, cs, additive))


def test_merkle_zk_and_errors(gpu):
    oracles = [rand_elems(5, 64, W)]
    salts = np.random.default_rng(3).integers(0, 256, size=(32, 32), dtype=np.uint8)
    assert np.array_equal(gpu.merkle_tree(oracles, 2, 0, salts), oracle.merkle_build(oracles, 2, True, salts))
    with pytest.raises(ValueError):
        gpu.merkle_tree([rand_elems(1, 2, W)], 2)
    with pytest.raises(ValueError):
        gpu.additive_FFT(rand_elems(1, 9, W), oracle.standard_basis(3, W), np.zeros(W, dtype=np.uint64))


def test_merkle_zk_2p18_leaves(gpu):
    # test_merkle_tree.cpp:106-115: zk tree with 2^18 leaves (the reference's salt batching case)
    L = 1 << 18
    oracles = [rand_elems(21, L, W), rand_elems(22, L, W)]
    salts = np.random.default_rng(4).integers(0, 256, size=(L, 32), dtype=np.uint8)
    assert np.array_equal(gpu.merkle_tree(oracles, 1, 0, salts), oracle.merkle_build(oracles, 1, True, salts))


# ---- full-size properties (BASELINE.json configs 2 / 3: sizes the oracle cannot sweep in seconds) --------
def _horner_at(coeffs, x):
    acc = np.zeros((1, W), dtype=np.uint64)
    # Horner in chunks through the oracle's elementwise multiply would be slow; evaluate via powers instead
    n = coeffs.shape[0]
    pw = np.zeros((n, W), dtype=np.uint64)
    pw[0, 0] = 1
    cur = 1
    xs = x[None, :]
    step = xs.copy()
    while cur < n:                      # pw[cur + i] = pw[i] * x^cur
        k = min(cur, n - cur)
        pw[cur:cur + k] = oracle.gf_mul(pw[:k], np.repeat(step, k, axis=0))
        step = oracle.gf_mul(step, step)
        cur *= 2
    prod = oracle.gf_mul(coeffs, pw)
    return np.bitwise_xor.reduce(prod, axis=0)


def test_cfg2_full_size_properties(gpu):
    m = 22
    basis, shift = _dom(m, "std0", 0)
    coeffs = rand_elems(0x2201, 1 << m, W)
    ev = gpu.additive_FFT(coeffs, basis, shift)
    # (1) spot values against Horner at the reference's point order (utils.tcc:8-30): index i <-> element i
    for idx in (0, 1, 2, 12345, (1 << m) - 1, 1 << 21):
        pt = np.array([idx, 0, 0], dtype=np.uint64)
        assert np.array_equal(ev[idx], _horner_at(coeffs, pt)), idx
    # (2) interpolate(evaluate) == id
    assert np.array_equal(gpu.additive_IFFT(ev, basis, shift), coeffs)
    # (3) GF(2)-linearity
    c2 = rand_elems(7, 1 << m, W)
    ev2 = gpu.additive_FFT(c2, basis, shift)
    assert np.array_equal(gpu.additive_FFT(coeffs ^ c2, basis, shift), ev ^ ev2)


def test_cfg3_fold_preserves_low_degree_at_full_size(gpu):
    # degree-2^20 codeword over 2^22 points, first two FRI rounds (eta = 1, 2): the folded word must again be
    # a codeword of the halved / quartered degree over the derived domain (checked by GPU IFFT: high coeffs zero)
    m, d = 22, 20
    basis, shift = _dom(m, "std0", 0)
    cw = gpu.additive_FFT(rand_elems(0x2203, 1 << d, W), basis, shift)
    loc = [1, 2]
    doms = oracle.fri_domains_additive(basis, shift, loc)
    cb, cs_, deg = basis, shift, 1 << d
    for i, eta in enumerate(loc):
        x = rand_elems(50 + i, 1, W)[0]
        cw = gpu.evaluate_next_f_i_over_entire_domain(cw, cb, cs_, 1 << eta, x)
        cb, cs_ = doms[i]
        deg >>= eta
        co = gpu.additive_IFFT(cw, cb, cs_)
        assert not co[deg:].any()


# ---- multiplicative cosets over the 181-bit prime field (BASELINE config 5's field) ----------------------------
def _fp_shifts():
    return [oracle.fp_one(), oracle.fp_from_ints([19])[0], oracle.fp_rand(5, 1)[0]]


@pytest.mark.parametrize("logn", [1, 2, 3, 6, 9, 12, 13, 16])
def test_mult_fft(gpu, logn):
    n = 1 << logn
    for ncoef in sorted({1, 2, 3, n // 2 + 1, n - 1, n}):
        if ncoef > n or ncoef < 1:
            continue
        coeffs = oracle.fp_rand(logn * 10 + ncoef, ncoef)
        for shift in _fp_shifts()[: (3 if logn < 12 else 2)]:
            assert np.array_equal(gpu.multiplicative_FFT(coeffs, logn, shift), oracle.multiplicative_fft(coeffs, n, shift)), (logn, ncoef)


def test_mult_fft_lde_shapes(gpu):
    shift = oracle.fp_from_ints([19])[0]            # Fractal's codeword coset shift = multiplicative_generator
    for logn, ncoef in [(14, 16), (18, 1 << 13), (18, (1 << 13) - 5), (20, 1 << 15)]:
        coeffs = oracle.fp_rand(ncoef, ncoef)
        assert np.array_equal(gpu.multiplicative_FFT(coeffs, logn, shift), oracle.multiplicative_fft(coeffs, 1 << logn, shift))
    assert not gpu.multiplicative_FFT(np.zeros((0, 3), dtype=np.uint64), 5, shift).any()


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 12, 13, 17])
def test_mult_ifft(gpu, logn):
    ev = oracle.fp_rand(logn, 1 << logn)
    for shift in _fp_shifts()[:2]:
        assert np.array_equal(gpu.multiplicative_IFFT(ev, shift), oracle.multiplicative_ifft(ev, shift))


def test_mult_ifft_known_degree(gpu):
    n, deg = 1 << 15, 3000
    shift = oracle.fp_from_ints([19])[0]
    ev = oracle.multiplicative_fft(oracle.fp_rand(1, deg), n, shift)
    assert np.array_equal(gpu.multiplicative_IFFT_of_known_degree(ev, deg, shift), oracle.multiplicative_ifft_known_degree(ev, deg, shift))


@pytest.mark.parametrize("logn,cs", [(1, 2), (4, 2), (6, 4), (8, 8), (14, 2), (14, 4), (5, 1)])
def test_mult_fri_fold(gpu, logn, cs):
    f = oracle.fp_rand(logn + cs, 1 << logn)
    x = oracle.fp_rand(99, 1)[0]
    for shift in _fp_shifts()[:2]:
        assert np.array_equal(gpu.multiplicative_evaluate_next_f_i(f, shift, cs, x), oracle.fri_fold_multiplicative(f, shift, cs, x))


def test_cfg5_shape_properties(gpu):
    # 2^22-point coset (Fractal codeword coset shift), degree-2^19 polynomial: round trip through the strided
    # IFFT of known degree, and the fold keeps the codeword low-degree over the squared coset
    logn, d = 22, 19
    shift = oracle.fp_from_ints([19])[0]
    coeffs = oracle.fp_rand(0x2205, 1 << d)
    cw = gpu.multiplicative_FFT(coeffs, logn, shift)
    assert np.array_equal(gpu.multiplicative_IFFT_of_known_degree(cw, 1 << d, shift), coeffs)
    x = oracle.fp_rand(7, 1)[0]
    nxt = gpu.multiplicative_evaluate_next_f_i(cw, shift, 4, x)
    shift4 = oracle.fp_mul(oracle.fp_mul(shift[None, :], shift[None, :]), oracle.fp_mul(shift[None, :], shift[None, :]))[0]
    co = gpu.multiplicative_IFFT(nxt, shift4)
    assert not co[1 << (d - 2):].any()


def test_sharded_transform_building_blocks(gpu):
    import dist_blocks_check as c
    c.check_pow_table(gpu, 5000)
    for log_n in (1, 2, 5, 10, 12, 15):
        c.check_taylor(gpu, log_n)
    c.check_combine(gpu, count=20000)


# ---- Poseidon over alt_bn128 Fr (algebraic leaf / two-to-one hashes of the BCS Merkle tree) --------------
import poseidon_cases as pc


def test_poseidon_to_montgomery(gpu):
    pc.check_to_montgomery(gpu)


def test_poseidon_permutation_kats(gpu):
    pc.check_permutation_kats(gpu)


@pytest.mark.parametrize("name", pc.SET_NAMES)
def test_poseidon_permutation(gpu, name):
    pc.check_permutation(gpu, name, count=200)


@pytest.mark.parametrize("name,r,cs,L,additive,zk", [
    ("test_params", 1, 1, 2, False, False), ("test_params", 1, 2, 8, False, True), ("test_params", 3, 2, 4, True, False),
    ("starkware_alpha5_t3", 2, 4, 64, False, False), ("high_alpha17_t3", 1, 2, 512, False, True), ("high_alpha17_t4", 2, 3, 128, False, True),
    ("high_alpha17_t4", 1, 6, 2, True, False), ("high_alpha17_t3", 4, 2, 4096, False, False), ("starkware_alpha5_t3", 1, 2, 2048, True, True),
    # 2^16 leaves: the first inner level (2^15 nodes) runs one lane per node, the rest one permutation over t lanes
    ("starkware_alpha5_t3", 1, 1, 1 << 16, False, False), ("high_alpha17_t4", 1, 1, 1 << 16, False, False),
])
def test_poseidon_merkle(gpu, name, r, cs, L, additive, zk):
    pc.check_merkle(gpu, name, r, cs, L, additive, zk)


def test_poseidon_leaf_and_two_to_one_kats(gpu):
    pc.check_leaf_and_two_to_one_kats(gpu)


def test_poseidon_errors(gpu):
    pc.check_errors(gpu)


# ---- proof of work (pow.tcc) -------------------------------------------------------------------------------
import pow_cases as pw


def test_pow_blake2b(gpu):
    pw.check_blake2b(gpu, [0, 1, 4, 9, 12, 17, 20], [1, 2, 3])


def test_pow_blake2b_reference_test_inputs(gpu):
    pw.check_blake2b_reference_test_inputs(gpu)
    pw.check_challenge_itself_passes(gpu)


@pytest.mark.parametrize("name", pc.SET_NAMES)
def test_pow_poseidon(gpu, name):
    pw.check_poseidon(gpu, name, [0, 3, 8, 13], [1, 2])


def test_pow_errors(gpu):
    pw.check_errors(gpu)


def test_pow_search_in_two_halves(gpu):
    pw.check_search_in_two_halves(gpu)


# ---- LDT reducer (ldt_reducer_aux.tcc:39-131) -----------------------------------------------------------------
import ldt_cases as lc


@pytest.mark.parametrize("m,degrees,seed,kind", lc.ADDITIVE + [(14, [1 << 12, 1 << 11, (1 << 12) - 5, 77], 6, "standard")])
def test_ldt_combine_additive(gpu, m, degrees, seed, kind):
    lc.check_additive(gpu, m, degrees, seed, kind)


@pytest.mark.parametrize("log_n,degrees,seed,shifted", lc.MULTIPLICATIVE + [(16, [1 << 14, (1 << 14) - 1, 12345], 5, True)])
def test_ldt_combine_multiplicative(gpu, log_n, degrees, seed, shifted):
    lc.check_multiplicative(gpu, log_n, degrees, seed, shifted)


def test_ldt_combine_full_size_sampled(gpu):
    # cfg3-sized codeword domain (2^22), Aurora-like degree spread
    lc.check_additive_sampled(gpu, 22, [(1 << 21) - 1, 1 << 20, (1 << 20) + 2 * 41 - 1, (1 << 21) - 1], 11)


@pytest.mark.parametrize("m,degrees,seed,kind", lc.GAP1)
def test_ldt_combine_gap1_group(gpu, m, degrees, seed, kind):
    lc.check_additive(gpu, m, degrees, seed, kind)


def test_ldt_combine_errors(gpu):
    lc.check_errors(gpu)


# ---- transcript extraction (merkle_tree.tcc:242-336, bcs_prover.tcc:187-197) ----------------------------------
import transcript_cases as tc


def test_membership_proofs(gpu):
    tc.check_membership_proofs(gpu, 1 << 12, 1, [1, 2, 5, 17, 64, 200, 5000])
    tc.check_membership_proofs(gpu, 2, 2, [1, 2, 3])
    tc.check_all_subsets_of_small_tree(gpu)
    tc.check_empty_and_errors(gpu)
    tc.check_deferred_downloads(gpu)
    tc.check_large_odd_deferred_downloads(gpu)


def test_query_responses(gpu):
    tc.check_query_responses(gpu, 1 << 16, 4, 5)
    tc.check_query_responses(gpu, 1 << 14, 2, 6, num_positions=400)
    tc.check_query_responses(gpu, 1 << 10, 17, 7)
    tc.check_wide_tree(gpu)


# ---- FRI-only SNARK (config 3's shape): device transcript == the oracle prover's; the oracle's verifier accepts / rejects ----
import fri_cases as fc


@pytest.mark.parametrize("field_name,dim,rs_extra,loc_param,interactions,queries", [
    ("gf192", 10, 3, 2, 1, 10), ("gf192", 16, 2, 2, 1, 10), ("gf192", 13, 2, 2, 2, 6),
    ("edwards_Fr", 10, 3, 2, 1, 10), ("edwards_Fr", 16, 2, 2, 1, 10)])
def test_fri_snark(gpu, field_name, dim, rs_extra, loc_param, interactions, queries):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    assert fc.prove_and_verify(gpu, torch, torch.device("cuda:0"), field_name, dim, rs_extra, loc_param, interactions, queries, 0x2203)


@pytest.mark.parametrize("field_name,dim,rs_extra,loc_param,interactions,queries", [("gf192", 16, 2, 2, 1, 10), ("gf192", 13, 2, 2, 2, 6), ("edwards_Fr", 16, 2, 2, 1, 10)])
def test_native_fri_snark(gpu, field_name, dim, rs_extra, loc_param, interactions, queries):
    """FRI_snark_prover through the C ABI (libiop_amd/cpp/fri.hpp inside the library) == the oracle prover's transcript."""
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    assert fc.native_prove_equals_oracle(gpu, torch, torch.device("cuda:0"), field_name, dim, rs_extra, loc_param, interactions, queries, 0x2203)


# ---- R1CS row check (rowcheck.tcc:16-88) -------------------------------------------------------------------------
@pytest.mark.parametrize("m,h,seed,kind", [(5, 2, 1, "aurora"), (8, 3, 2, "general"), (7, 7, 3, "aurora"), (6, 0, 4, "general"), (16, 11, 5, "aurora")])
def test_rowcheck_additive(gpu, m, h, seed, kind):
    lc.check_rowcheck_additive(gpu, m, h, seed, kind)


def test_rowcheck_is_a_polynomial_division(gpu):
    lc.check_rowcheck_is_a_polynomial_division(gpu, 14, 9, 3)


@pytest.mark.parametrize("log_n,log_h,seed", [(5, 2, 1), (8, 4, 2), (6, 6, 3), (7, 0, 4), (15, 10, 5)])
def test_rowcheck_multiplicative(gpu, log_n, log_h, seed):
    lc.check_rowcheck_multiplicative(gpu, log_n, log_h, seed)


def test_rowcheck_errors(gpu):
    lc.check_rowcheck_errors(gpu)


# ---- fz virtual oracle (r1cs_rs_iop.tcc:181-222) ---------------------------------------------------------------------
@pytest.mark.parametrize("m,idim,seed,kind", [(5, 2, 1, "aurora"), (9, 4, 2, "general"), (7, 0, 3, "general"), (16, 5, 4, "aurora")])
def test_fz_additive(gpu, m, idim, seed, kind):
    lc.check_fz_additive(gpu, m, idim, seed, kind)


@pytest.mark.parametrize("log_n,ilog,seed", [(5, 2, 1), (9, 4, 2), (14, 3, 3), (6, 0, 4)])
def test_fz_multiplicative(gpu, log_n, ilog, seed):
    lc.check_fz_multiplicative(gpu, log_n, ilog, seed)


# ---- sumcheck g oracle (sumcheck.tcc:58-119) -------------------------------------------------------------------------
@pytest.mark.parametrize("m,sdim,seed,kind", [(5, 2, 1, "aurora"), (9, 4, 2, "general"), (7, 3, 3, "unshifted"), (16, 11, 4, "aurora"), (3, 1, 5, "general")])
def test_sumcheck_g_additive(gpu, m, sdim, seed, kind):
    lc.check_sumcheck_g_additive(gpu, m, sdim, seed, kind)


@pytest.mark.parametrize("log_n,slog,seed", [(5, 2, 1), (9, 4, 2), (15, 10, 3), (6, 0, 4)])
def test_sumcheck_g_multiplicative(gpu, log_n, slog, seed):
    lc.check_sumcheck_g_multiplicative(gpu, log_n, slog, seed)


@pytest.mark.parametrize("n,k,seed,prime", [(16, 3, 1, False), (300, 1, 2, False), (1 << 16, 3, 3, False), (1 << 15, 3, 4, True), (7, 2, 5, True)])
def test_lincheck(gpu, n, k, seed, prime):
    lc.check_lincheck(gpu, n, k, seed, prime)


# ---- batched transforms over one domain ---------------------------------------------------------------------------------
@pytest.mark.parametrize("m,batch", [(1, 3), (9, 4), (14, 3), (18, 2)])
def test_batched_ifft(gpu, m, batch):
    basis, shift = rand_elems(40 + m, m, W), rand_elems(41 + m, 1, W)[0]
    ev = rand_elems(42 + m, batch << m, W)
    d_in, d_out = gpu.malloc(ev.nbytes), gpu.malloc(ev.nbytes)
    try:
        gpu.h2d(d_in, ev)
        gpu.additive_IFFT_batch_dev(d_in, batch, basis, shift, d_out)
        out = np.empty_like(ev)
        gpu.d2h(out, d_out)
    finally:
        gpu.free(d_in)
        gpu.free(d_out)
    for k in range(batch):
        assert np.array_equal(out[k << m:(k + 1) << m], oracle.additive_ifft(ev[k << m:(k + 1) << m], basis, shift)), k


@pytest.mark.parametrize("m,ncoef,batch,cb,cc", [(8, 50, 3, 0, 4), (10, 64, 2, 3, 5), (16, 1 << 12, 4, 1, 2), (6, 1, 2, 0, 64)])
def test_batched_lde(gpu, m, ncoef, batch, cb, cc):
    basis, shift = rand_elems(50 + m, m, W), rand_elems(51 + m, 1, W)[0]
    d = 0 if ncoef <= 1 else int(ncoef - 1).bit_length()
    polys = [rand_elems(60 + k, ncoef, W) for k in range(batch)]
    ins = [gpu.malloc(p.nbytes) for p in polys]
    outs = [gpu.malloc((cc << d) * 24) for _ in polys]
    try:
        for b, p in zip(ins, polys):
            gpu.h2d(b, p)
        gpu.additive_LDE_batch_dev(ins, ncoef, basis, shift, cb, cc, outs)
        for k in range(batch):
            got = np.empty((cc << d, W), dtype=np.uint64)
            gpu.d2h(got, outs[k])
            assert np.array_equal(got, oracle.additive_fft(polys[k], basis, shift)[cb << d:(cb + cc) << d]), k
    finally:
        for b in ins + outs:
            gpu.free(b)


# ---- encoded Aurora prover: device transcript == the oracle prover's, byte for byte (configs 1 and 4's shapes) ------------------
import aurora_cases as ac


@pytest.mark.parametrize("field_name,log_n,num_inputs", [("gf192", 8, 15), ("gf192", 10, 15), ("gf192", 12, 15),      # cfg4's shape: RS 5, localization 2
                                                         ("edwards_Fr", 8, 15), ("edwards_Fr", 12, 15)])               # cfg1: 2^12 over the 181-bit field
def test_aurora_transcript_equals_oracle_prover(gpu, field_name, log_n, num_inputs):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    transcript, params = ac.check_transcript_equals_oracle(gpu, torch, torch.device("cuda:0"), field_name, log_n, num_inputs, 0x2204)
    assert params.RS_extra_dimensions == 5 and params.localization_parameters[:2] == [1, 2]
    if log_n == 12:
        code = ac.FIELDS[field_name][0]
        for label, data in ac.tamper_cases(transcript):
            assert not oracle.aurora_verify(code, log_n, num_inputs, 0x2204, data), label


@pytest.mark.parametrize("field_name,log_n,num_inputs", [("gf192", 8, 15), ("gf192", 12, 15), ("edwards_Fr", 9, 7), ("edwards_Fr", 12, 15)])
def test_native_aurora_prover_behind_the_c_abi(gpu, field_name, log_n, num_inputs):
    """iopx_aurora_prove — the C++ prover surface (libiop_amd/cpp/aurora.hpp) inside the library — against the oracle prover, byte for byte."""
    n = 1 << log_n
    inst = gpu.aurora_example_instance({"gf192": 0, "edwards_Fr": 1}[field_name], n, num_inputs, n - 1, 0x2204)
    try:
        ref = oracle.aurora_prove(ac.FIELDS[field_name][0], log_n, num_inputs, 0x2204)
        assert gpu.aurora_prove(inst) == ref
        assert gpu.aurora_prove(inst) == ref
    finally:
        gpu.aurora_instance_free(inst)


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_aurora_other_rates(gpu, field_name):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    ac.check_transcript_equals_oracle(gpu, torch, torch.device("cuda:0"), field_name, 9, 7, 5, rs_extra=3, localization=3)


@pytest.mark.parametrize("m,d,batch", [(9, 5, 3), (14, 9, 2), (18, 14, 3), (19, 16, 1)])
def test_reextend_equals_ifft_then_fft(gpu, m, d, batch):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    tc.check_reextend(gpu, torch, torch.device("cuda:0"), m, d, batch, 60 + m)


# ---- Fractal indexer and prover: device index root and transcript == the oracle's, byte for byte (config 5's shape) ---------------
import fractal_cases as frc


@pytest.mark.parametrize("field_name,log_n,num_inputs", [("edwards_Fr", 8, 0), ("edwards_Fr", 10, 0), ("edwards_Fr", 12, 0),   # cfg5: k = 0, RS 3, localization 2
                                                         ("edwards_Fr", 9, 15), ("gf192", 7, 15), ("gf192", 9, 15)])
def test_fractal_transcript_equals_oracle_prover(gpu, field_name, log_n, num_inputs):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    transcript, roots, params = frc.check_transcript_equals_oracle(gpu, torch, torch.device("cuda:0"), field_name, log_n, num_inputs, 0x2205)
    assert params.RS_extra_dimensions == 3 and params.localization_parameters[:2] == [1, 2] and len(roots) == 1
    if log_n == 12:
        code = frc.FIELDS[field_name][0]
        for label, data in frc.tamper_cases(transcript):
            assert not oracle.fractal_verify(code, log_n, num_inputs, 0x2205, data, roots), label


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_fractal_index_oracles_and_other_rates(gpu, field_name):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    frc.check_index_oracles(gpu, torch, torch.device("cuda:0"), field_name, 6, 3, 0x2205)
    frc.check_transcript_equals_oracle(gpu, torch, torch.device("cuda:0"), field_name, 7, 3, 5, rs_extra=2, localization=3)


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
@pytest.mark.parametrize("n", [1, 300, 70000, 1 << 20])
def test_fractal_div_kernel(gpu, field_name, n):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    if n > 100000 and field_name == "edwards_Fr":
        n = 200000                                   # the host side builds its inputs with Python integers
    frc.check_div_kernel(gpu, torch, torch.device("cuda:0"), field_name, n)


@pytest.mark.parametrize("field_name,log_l,log_h", [("gf192", 9, 4), ("gf192", 18, 12), ("edwards_Fr", 9, 4), ("edwards_Fr", 18, 12)])
def test_fractal_domain_kernels(gpu, field_name, log_l, log_h):
    import torch
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    frc.check_domain_kernels(gpu, torch, torch.device("cuda:0"), field_name, log_l, log_h, samples=(0, 1, 255, 256, 4095, 4096, 70001))


def test_half_wavefront_product_keeps_exec(gpu):
    """gf_mul_halves narrows EXEC to one half of the wavefront at a time inside its asm block and restores it (the block no longer names EXEC on its
    clobber list): under a divergent branch the products are right, untaken lanes keep their input, and a ballot right after the product sees
    exactly the lanes that took the branch."""
    import halves_cases
    halves_cases.check(gpu, None, count_active=True)


def test_membership_proofs_validate_with_hashlib_only(gpu):
    """tests/bcs/test_merkle_tree.cpp:117-167 on the HIP library: an 8-leaf tree over two oracles and the pruned multi-membership proof of EVERY subset of its
    leaves, validated by a pure-Python verifier over hashlib (tests/test_independent_pins.py) — no oracle code involved."""
    import test_independent_pins as pins
    pins.check_every_subset_validates(gpu)


def test_noncanonical_prime_field_inputs_give_canonical_congruent_outputs(gpu):
    import noncanonical_cases
    noncanonical_cases.check(gpu)
