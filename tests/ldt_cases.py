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


# degree gap 1 over one-word points (standard basis, one-word shift): the group takes two comb products per oracle and one one-word product
# (k_ldt_combine_add_slots small_slot); gap-1 oracles alone, next to other gaps, next to maximal ones, and with a random basis (general path)
GAP1 = [
    (8, [100, 99], 31, "standard"), (9, [300, 299, 299, 299], 32, "standard"), (10, [512, 511, 496, 511, 512, 512, 511], 33, "standard"),
    (7, [64, 63, 62, 63], 34, "standard"), (8, [200, 199, 199], 35, "random"), (12, [4096, 4095, 4080, 4096, 4096, 4096, 4095], 36, "standard"),
]

ADDITIVE = [
    (1, [1], 0, "random"), (3, [5, 5, 5], 1, "random"), (6, [40, 17, 40, 33], 2, "random"), (9, [300, 44, 1, 299, 300], 3, "standard"),
    (10, [1 << 9, (1 << 9) - 1, 7], 4, "standard"), (4, [3, 9], 5, "random"),
    # Aurora-like spread: exponents 2^k - 1 (shared by three oracles), 2^k - 82 (shares 17 bits with it), 1, 2^k
    (8, [(1 << 7) - 1, 1 << 6, 1 << 6, 1 << 6, (1 << 6) + 21, (1 << 7) - 2, (1 << 6) - 1], 6, "standard"),
    # more distinct exponents than shared-power slots: the direct kernel
    (5, [200] + [200 - 3 * k - 1 for k in range(20)], 7, "random"),
    # all multi-bit exponents identical to the common set
    (6, [100, 100 - 7, 100 - 7, 100 - 15], 8, "random"),
    # 11..16 distinct degree gaps: the shared-power slots need more than 64 KiB of LDS (hipFuncSetAttribute path)
    (9, [400] + [400 - (1 << k) for k in range(8)] + [400 - 3, 400 - 5, 400 - 9, 400 - 17, 400 - 33], 9, "standard"),
    (9, [400] + [400 - 3 * k - 1 for k in range(15)], 10, "random"),
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


# ---- R1CS row check (rowcheck.tcc:16-88) -------------------------------------------------------------------------------
def check_rowcheck_additive(lib, m, h, seed, kind="aurora"):
    n = 1 << m
    if kind == "aurora":        # standard basis, codeword domain shifted off the constraint domain (aurora_iop.tcc:37-43)
        basis, shift, cshift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64), np.zeros(W, dtype=np.uint64)
    else:
        basis, shift, cshift = rand_elems(seed + 1, m, W), rand_elems(seed + 2, 1, W)[0], rand_elems(seed + 3, 1, W)[0]
    az, bz, cz = rand_elems(seed + 4, n, W), rand_elems(seed + 5, n, W), rand_elems(seed + 6, n, W)
    got = lib.rowcheck(az, bz, cz, basis, shift, h, cshift)
    assert np.array_equal(got, oracle.rowcheck_additive(az, bz, cz, basis, shift, h, cshift))


def check_rowcheck_is_a_polynomial_division(lib, m, h, seed):
    """When Az Bz - Cz vanishes on H the result is the codeword of the quotient polynomial: interpolating it gives degree
    < 2 deg - |H| (the identity the protocol rests on, r1cs_rs_iop.tcc:372-377)."""
    n, dH = 1 << m, 1 << h
    basis, shift, cshift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64), np.zeros(W, dtype=np.uint64)
    a_c, b_c = rand_elems(seed, dH, W), rand_elems(seed + 1, dH, W)
    hb = basis[:h]
    a_on_h, b_on_h = oracle.additive_fft(a_c, hb, cshift), oracle.additive_fft(b_c, hb, cshift)
    c_c = oracle.additive_ifft(oracle.gf_mul(a_on_h, b_on_h), hb, cshift)       # C = A B on H, degree < |H|
    az, bz, cz = (oracle.additive_fft(v, basis, shift) for v in (a_c, b_c, c_c))
    q = lib.rowcheck(az, bz, cz, basis, shift, h, cshift)
    coeffs = oracle.additive_ifft(q, basis, shift)
    assert not coeffs[dH - 1:].any()            # deg(A B - C) <= 2 |H| - 2, so the quotient has degree <= |H| - 2


def check_rowcheck_multiplicative(lib, log_n, log_h, seed):
    n = 1 << log_n
    gen = libiop_amd.edwards_subgroup_generator(log_n)
    shift = libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR])[0]
    cshift = libiop_amd.edwards_to_montgomery([1])[0]
    az, bz, cz = _rand_fp(seed, n), _rand_fp(seed + 1, n), _rand_fp(seed + 2, n)
    got = lib.rowcheck_multiplicative(az, bz, cz, log_n, gen, shift, log_h, cshift)
    assert np.array_equal(got, oracle.rowcheck_fp(az, bz, cz, shift, 1 << log_h, cshift))


def check_rowcheck_errors(lib):
    basis, zero = oracle.standard_basis(4, W), np.zeros(W, dtype=np.uint64)
    v = rand_elems(1, 16, W)
    with pytest.raises(ValueError):         # L = H-aligned and unshifted: Z_H vanishes on the codeword domain
        lib.rowcheck(v, v, v, basis, zero, 2, zero)


# ---- fz virtual oracle (r1cs_rs_iop.tcc:181-222) -----------------------------------------------------------------------
def check_fz_additive(lib, m, idim, seed, kind="aurora"):
    n = 1 << m
    if kind == "aurora":
        basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
        ib, ish = oracle.standard_basis(idim, W) if idim else np.zeros((0, W), dtype=np.uint64), np.zeros(W, dtype=np.uint64)
    else:
        basis, shift = rand_elems(seed + 1, m, W), rand_elems(seed + 2, 1, W)[0]
        ib, ish = rand_elems(seed + 3, max(idim, 1), W)[:idim], rand_elems(seed + 4, 1, W)[0]
    fw, f1v = rand_elems(seed + 5, n, W), rand_elems(seed + 6, n, W)
    got = lib.fz(fw, f1v, basis, shift, ib, ish)
    assert np.array_equal(got, oracle.fz_additive(fw, f1v, basis, shift, ib, ish))


def check_fz_multiplicative(lib, log_n, ilog, seed):
    n = 1 << log_n
    gen = libiop_amd.edwards_subgroup_generator(log_n)
    shift = libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR])[0]
    ish = libiop_amd.edwards_to_montgomery([1])[0]
    fw, f1v = _rand_fp(seed, n), _rand_fp(seed + 1, n)
    got = lib.fz_multiplicative(fw, f1v, log_n, gen, shift, ilog, ish)
    assert np.array_equal(got, oracle.fz_fp(fw, f1v, shift, 1 << ilog, ish))


# ---- sumcheck g oracle (sumcheck.tcc:58-119) ---------------------------------------------------------------------------
def check_sumcheck_g_additive(lib, m, sdim, seed, kind="aurora"):
    n = 1 << m
    if kind == "aurora":
        basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
        sb, ssh = oracle.standard_basis(sdim, W) if sdim else np.zeros((0, W), dtype=np.uint64), np.zeros(W, dtype=np.uint64)
    elif kind == "unshifted":       # the codeword domain contains zero: its inverse is taken as zero (utils.tcc:79-97)
        basis, shift = rand_elems(seed + 1, m, W), np.zeros(W, dtype=np.uint64)
        sb, ssh = rand_elems(seed + 3, max(sdim, 1), W)[:sdim], rand_elems(seed + 4, 1, W)[0]
    else:
        basis, shift = rand_elems(seed + 1, m, W), rand_elems(seed + 2, 1, W)[0]
        sb, ssh = rand_elems(seed + 3, max(sdim, 1), W)[:sdim], rand_elems(seed + 4, 1, W)[0]
    f, h, mu = rand_elems(seed + 5, n, W), rand_elems(seed + 6, n, W), rand_elems(seed + 7, 1, W)[0]
    got = lib.sumcheck_g(f, h, basis, shift, sb, ssh, mu)
    assert np.array_equal(got, oracle.sumcheck_g_additive(f, h, basis, shift, sb, ssh, mu))
    zero = np.zeros(W, dtype=np.uint64)     # the claimed sum of every lincheck instance: the inversion-free kernel
    assert np.array_equal(lib.sumcheck_g(f, h, basis, shift, sb, ssh, zero), oracle.sumcheck_g_additive(f, h, basis, shift, sb, ssh, zero))


def check_sumcheck_g_multiplicative(lib, log_n, slog, seed):
    n = 1 << log_n
    gen = libiop_amd.edwards_subgroup_generator(log_n)
    shift = libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR])[0]
    ssh = libiop_amd.edwards_to_montgomery([1])[0]
    f, h, mu = _rand_fp(seed, n), _rand_fp(seed + 1, n), _rand_fp(seed + 2, 1)[0]
    got = lib.sumcheck_g_multiplicative(f, h, log_n, gen, shift, slog, ssh, mu)
    assert np.array_equal(got, oracle.sumcheck_g_fp(f, h, shift, 1 << slog, ssh, mu))


# ---- lincheck virtual oracle (basic_lincheck_aux.tcc:102-144) -----------------------------------------------------------
def check_lincheck(lib, n, k, seed, prime_field=False):
    gen = (lambda s, c: _rand_fp(s, c)) if prime_field else (lambda s, c: rand_elems(s, c, W))
    fz, p1, p2 = gen(seed, n), gen(seed + 1, n), gen(seed + 2, n)
    mz = [gen(seed + 10 + m, n) for m in range(k)]
    r = gen(seed + 3, k)
    got = lib.lincheck(fz, mz, r, p1, p2, prime_field)
    assert np.array_equal(got, oracle.lincheck_combine(fz, mz, r, p1, p2, prime_field))
    with pytest.raises(ValueError):         # basic_lincheck_aux.tcc:32-34
        lib.lincheck(fz, mz, r[:k - 1] if k > 1 else np.zeros((2, 3), dtype=np.uint64), p1, p2, prime_field)
