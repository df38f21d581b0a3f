"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper over ``oracle/liboracle.so`` (the CPU restatement of the reference algorithms, see the
headers of ``oracle/*.hpp``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import this package; the product (``libiop_amd``) never does.

Field elements travel as ``numpy.uint64`` arrays of shape ``(count, words)`` — the raw little-endian
word layout of libff's binary fields (gf64: 1 word, gf128: 2, gf192: 3, gf256: 4).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_szp = ctypes.POINTER(ctypes.c_size_t)


def build(force=False):
    """Compile oracle/liboracle.so with the Makefile next to this file."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp")) or f == "Makefile"]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_localization_array.restype = ctypes.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p)


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _words(a):
    return int(a.shape[-1])


def has_pclmul():
    return bool(lib().oracle_has_pclmul())


def clmul64(a, b, portable=False):
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    f = lib().oracle_clmul64_portable if portable else lib().oracle_clmul64
    f(ctypes.c_uint64(a), ctypes.c_uint64(b), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def gf_mul(a, b):
    a, b = _c(a), _c(b)
    out = np.empty_like(a)
    rc = lib().oracle_gf_mul(_words(a), _p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    assert rc == 0
    return out


def gf_inv(a):
    a = _c(a)
    out = np.empty_like(a)
    rc = lib().oracle_gf_inv(_words(a), _p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    assert rc == 0
    return out


def standard_basis(m, words):
    """libiop/algebra/field_subset/subspace.tcc:93-108 — basis element i is FieldT(1 << i)."""
    b = np.zeros((m, words), dtype=np.uint64)
    for i in range(m):
        b[i, 0] = np.uint64(1) << np.uint64(i)
    return b


def all_subset_sums(basis, shift):
    basis, shift = _c(basis), _c(shift)
    m, w = basis.shape
    out = np.empty((1 << m, w), dtype=np.uint64)
    rc = lib().oracle_all_subset_sums(w, _p(basis), ctypes.c_size_t(m), _p(shift), _p(out))
    assert rc == 0
    return out


def naive_fft(coeffs, basis, shift):
    coeffs, basis, shift = _c(coeffs), _c(basis), _c(shift)
    m, w = basis.shape
    out = np.empty((1 << m, w), dtype=np.uint64)
    rc = lib().oracle_naive_fft(w, _p(coeffs), ctypes.c_size_t(coeffs.shape[0]), _p(basis), ctypes.c_size_t(m), _p(shift), _p(out))
    assert rc == 0
    return out


def additive_fft(coeffs, basis, shift):
    coeffs, basis, shift = _c(coeffs), _c(basis), _c(shift)
    m, w = basis.shape
    out = np.empty((1 << m, w), dtype=np.uint64)
    rc = lib().oracle_additive_fft(w, _p(coeffs), ctypes.c_size_t(coeffs.shape[0]), _p(basis), ctypes.c_size_t(m), _p(shift), _p(out))
    if rc != 0:
        raise ValueError("oracle_additive_fft rc=%d" % rc)
    return out


def additive_ifft(evals, basis, shift):
    evals, basis, shift = _c(evals), _c(basis), _c(shift)
    m, w = basis.shape
    assert evals.shape[0] == 1 << m
    out = np.empty((1 << m, w), dtype=np.uint64)
    rc = lib().oracle_additive_ifft(w, _p(evals), _p(basis), ctypes.c_size_t(m), _p(shift), _p(out))
    assert rc == 0
    return out


def additive_ifft_known_degree(evals, degree, basis, shift):
    evals, basis, shift = _c(evals), _c(basis), _c(shift)
    m, w = basis.shape
    k = max(degree - 1, 0).bit_length()
    out = np.empty((1 << k, w), dtype=np.uint64)
    rc = lib().oracle_additive_ifft_known_degree(w, _p(evals), ctypes.c_size_t(degree), _p(basis), ctypes.c_size_t(m), _p(shift), _p(out))
    assert rc == 0
    return out


def fri_fold_additive(f_i, basis, shift, coset_size, x_i):
    f_i, basis, shift, x_i = _c(f_i), _c(basis), _c(shift), _c(x_i)
    m, w = basis.shape
    out = np.empty(((1 << m) // coset_size, w), dtype=np.uint64)
    rc = lib().oracle_fri_fold_additive(w, _p(f_i), _p(basis), ctypes.c_size_t(m), _p(shift), ctypes.c_size_t(coset_size), _p(x_i), _p(out))
    assert rc == 0
    return out


def fri_domains_additive(basis, shift, loc_params):
    """Returns [(basis_i, shift_i)] for L^(1).. following fri_ldt.tcc:310-338."""
    basis, shift = _c(basis), _c(shift)
    m, w = basis.shape
    dims, d = [], m
    for eta in loc_params:
        d -= eta
        dims.append(d)
    out_b = np.zeros((sum(dims), w), dtype=np.uint64)
    out_s = np.zeros((len(dims), w), dtype=np.uint64)
    loc = (ctypes.c_size_t * len(loc_params))(*loc_params)
    rc = lib().oracle_fri_domains_additive(w, _p(basis), ctypes.c_size_t(m), _p(shift), loc, ctypes.c_size_t(len(loc_params)), _p(out_b), _p(out_s))
    assert rc == 0
    res, off = [], 0
    for i, dd in enumerate(dims):
        res.append((out_b[off:off + dd].copy(), out_s[i].copy()))
        off += dd
    return res


def localization_array(loc_param, codeword_dim, rs_extra):
    buf = (ctypes.c_size_t * 64)()
    n = lib().oracle_localization_array(ctypes.c_size_t(loc_param), ctypes.c_size_t(codeword_dim), ctypes.c_size_t(rs_extra), buf, ctypes.c_size_t(64))
    return [int(buf[i]) for i in range(n)]


def next_coset_query_positions(additive, non_localized_n, localized_n, seed_position, prev_loc, cur_loc):
    buf = (ctypes.c_size_t * (1 << cur_loc))()
    lib().oracle_next_coset_query_positions(int(additive), ctypes.c_size_t(non_localized_n), ctypes.c_size_t(localized_n),
                                            ctypes.c_size_t(seed_position), ctypes.c_size_t(prev_loc), ctypes.c_size_t(cur_loc), buf)
    return [int(v) for v in buf]


def blake2b(data, outlen=32, key=b""):
    out = (ctypes.c_uint8 * outlen)()
    d = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) if len(data) else b"\0")
    k = (ctypes.c_uint8 * max(len(key), 1)).from_buffer_copy(bytes(key) if len(key) else b"\0")
    lib().oracle_blake2b(out, ctypes.c_size_t(outlen), d, ctypes.c_size_t(len(data)), k, ctypes.c_size_t(len(key)))
    return bytes(out)


def merkle_build(oracles, coset_size, additive=True, salts=None):
    """oracles: list of (n, words) uint64 arrays.  Returns (2L-1, 32) uint8 node array in heap order."""
    oracles = [_c(o) for o in oracles]
    n, w = oracles[0].shape
    L = n // coset_size
    nodes = np.zeros((2 * L - 1, 32), dtype=np.uint8)
    ptrs = (ctypes.c_void_p * len(oracles))(*[o.ctypes.data for o in oracles])
    if salts is not None:
        salts = np.ascontiguousarray(salts, dtype=np.uint8)
        sp, sb = salts.ctypes.data_as(_u8p), salts.shape[1]
    else:
        sp, sb = None, 0
    rc = lib().oracle_merkle_build(ptrs, ctypes.c_size_t(len(oracles)), ctypes.c_size_t(8 * w), ctypes.c_size_t(n),
                                   ctypes.c_size_t(coset_size), int(additive), sp, ctypes.c_size_t(sb),
                                   nodes.ctypes.data_as(_u8p))
    if rc != 0:
        raise ValueError("Merkle tree size must be a power of two, and at least 2.")
    return nodes


class Hashchain:
    """libiop/bcs/hashing/blake2b.tcc:10-110 restated (see oracle/merkle.hpp)."""

    def __init__(self):
        self.state = (ctypes.c_uint8 * 32)()
        self.idx = ctypes.c_uint64(0)
        lib().oracle_hashchain_init(self.state, ctypes.byref(self.idx))

    def absorb(self, digest):
        d = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(digest))
        lib().oracle_hashchain_absorb(self.state, d)

    def squeeze(self, num_elements, words):
        out = np.zeros((num_elements, words), dtype=np.uint64)
        lib().oracle_hashchain_squeeze_binary(self.state, ctypes.byref(self.idx), ctypes.c_size_t(num_elements),
                                              ctypes.c_size_t(8 * words), out.ctypes.data_as(_u8p))
        return out

    def squeeze_query_positions(self, num_positions, rng):
        buf = (ctypes.c_size_t * num_positions)()
        rc = lib().oracle_hashchain_squeeze_positions(self.state, ctypes.byref(self.idx), ctypes.c_size_t(num_positions), ctypes.c_size_t(rng), buf)
        if rc != 0:
            raise ValueError("upper_bound must be a power of two.")
        return [int(v) for v in buf]


# ---- prime field edwards_Fr (Montgomery words, (count, 3) uint64) and the multiplicative-domain path ----------
EDWARDS_R = 1552511030102430251236801561344621993261920897571225601


def fp_from_ints(values):
    """canonical Python ints -> Montgomery words"""
    c = np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(3)] for v in values], dtype=np.uint64).reshape(-1, 3)
    out = np.empty_like(c)
    lib().oracle_fp_from_canonical(_p(c), _p(out), ctypes.c_size_t(c.shape[0]))
    return out


def fp_to_ints(m):
    m = _c(m).reshape(-1, 3)
    out = np.empty_like(m)
    lib().oracle_fp_to_canonical(_p(m), _p(out), ctypes.c_size_t(m.shape[0]))
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) for r in out]


def fp_rand(seed, count):
    rng = np.random.Generator(np.random.PCG64(seed))
    vals = [int.from_bytes(rng.bytes(32), "little") % EDWARDS_R for _ in range(count)]
    return fp_from_ints(vals)


def _fp_bin(op, a, b):
    a, b = _c(a), _c(b)
    out = np.empty_like(a)
    lib().oracle_fp_binop(op, _p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp_mul(a, b):
    return _fp_bin(0, a, b)


def fp_add(a, b):
    return _fp_bin(1, a, b)


def fp_sub(a, b):
    return _fp_bin(2, a, b)


def fp_inv(a):
    a = _c(a)
    out = np.empty_like(a)
    lib().oracle_fp_inv(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fp_subgroup_generator(order):
    out = np.empty(3, dtype=np.uint64)
    lib().oracle_fp_subgroup_generator(ctypes.c_size_t(order), _p(out))
    return out


def fp_one():
    return fp_from_ints([1])[0]


def fp_all_elements(order, shift):
    shift = _c(shift)
    out = np.empty((order, 3), dtype=np.uint64)
    lib().oracle_fp_all_elements(ctypes.c_size_t(order), _p(shift), _p(out))
    return out


def fp_naive_fft(coeffs, order, shift):
    coeffs, shift = _c(coeffs), _c(shift)
    out = np.empty((order, 3), dtype=np.uint64)
    lib().oracle_fp_naive_fft(_p(coeffs), ctypes.c_size_t(coeffs.shape[0]), ctypes.c_size_t(order), _p(shift), _p(out))
    return out


def multiplicative_fft(coeffs, order, shift):
    coeffs, shift = _c(coeffs), _c(shift)
    out = np.empty((order, 3), dtype=np.uint64)
    rc = lib().oracle_fp_fft(_p(coeffs), ctypes.c_size_t(coeffs.shape[0]), ctypes.c_size_t(order), _p(shift), _p(out))
    if rc != 0:
        raise ValueError("oracle_fp_fft rc=%d" % rc)
    return out


def multiplicative_ifft(evals, shift):
    evals, shift = _c(evals), _c(shift)
    out = np.empty_like(evals)
    lib().oracle_fp_ifft(_p(evals), ctypes.c_size_t(evals.shape[0]), _p(shift), _p(out))
    return out


def multiplicative_ifft_known_degree(evals, degree, shift):
    evals, shift = _c(evals), _c(shift)
    k = max(degree - 1, 0).bit_length()
    out = np.empty((1 << k, 3), dtype=np.uint64)
    lib().oracle_fp_ifft_known_degree(_p(evals), ctypes.c_size_t(degree), ctypes.c_size_t(evals.shape[0]), _p(shift), _p(out))
    return out


def fri_fold_multiplicative(f_i, shift, coset_size, x_i):
    f_i, shift, x_i = _c(f_i), _c(shift), _c(x_i)
    out = np.empty((f_i.shape[0] // coset_size, 3), dtype=np.uint64)
    lib().oracle_fp_fri_fold(_p(f_i), ctypes.c_size_t(f_i.shape[0]), _p(shift), ctypes.c_size_t(coset_size), _p(x_i), _p(out))
    return out


# ---- Poseidon over alt_bn128 Fr ((count, 4) uint64 Montgomery words) -------------------------------------------------
BN128_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _ints_to_words4(values):
    return np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in values], dtype=np.uint64).reshape(-1, 4)


def bn_from_ints(values):
    c = _ints_to_words4(values)
    out = np.empty_like(c)
    lib().oracle_bn_from_canonical(_p(c), _p(out), ctypes.c_size_t(c.shape[0]))
    return out


def bn_to_ints(m):
    m = _c(m).reshape(-1, 4)
    out = np.empty_like(m)
    lib().oracle_bn_to_canonical(_p(m), _p(out), ctypes.c_size_t(m.shape[0]))
    return [sum(int(r[i]) << (64 * i) for i in range(4)) for r in out]


class PoseidonParams:
    """One parameter set (dict with alpha, full_rounds, partial_rounds, rate, state_size, near_mds, mds, ark as ints)."""

    def __init__(self, d):
        self.d = d
        self.mds = _ints_to_words4([v for row in d["mds"] for v in row])
        self.ark = _ints_to_words4([v for row in d["ark"] for v in row])

    def args(self):
        d = self.d
        return (ctypes.c_size_t(d["alpha"]), ctypes.c_size_t(d["full_rounds"]), ctypes.c_size_t(d["partial_rounds"]),
                ctypes.c_size_t(d["rate"]), ctypes.c_size_t(d["state_size"]), int(bool(d["near_mds"])), _p(self.mds), _p(self.ark))


def poseidon_permute(params, state):
    state = _c(state).copy()
    lib().oracle_poseidon_permute(*params.args(), _p(state))
    return state


def poseidon_leafhash(params, leaf):
    leaf = _c(leaf)
    out = np.empty(4, dtype=np.uint64)
    lib().oracle_poseidon_leafhash(*params.args(), _p(leaf), ctypes.c_size_t(leaf.shape[0]), _p(out))
    return out


def poseidon_two_to_one(params, l, r):
    l, r = _c(l), _c(r)
    out = np.empty(4, dtype=np.uint64)
    lib().oracle_poseidon_two_to_one(*params.args(), _p(l), _p(r), _p(out))
    return out


def poseidon_salt_to_field(salt):
    out = np.empty(4, dtype=np.uint64)
    buf = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(salt))
    lib().oracle_poseidon_salt_to_field(buf, _p(out))
    return out


def poseidon_merkle(params, oracles, coset_size, additive=False, salts=None):
    oracles = [_c(o) for o in oracles]
    n = oracles[0].shape[0]
    L = n // coset_size
    nodes = np.zeros((2 * L - 1, 4), dtype=np.uint64)
    if salts is not None:
        salts = np.ascontiguousarray(salts, dtype=np.uint8)
        assert salts.shape == (L, 32)
    ptrs = (ctypes.c_void_p * len(oracles))(*[o.ctypes.data for o in oracles])
    lib().oracle_poseidon_merkle(*params.args(), ptrs, ctypes.c_size_t(len(oracles)), ctypes.c_size_t(n), ctypes.c_size_t(coset_size),
                                 int(additive), ctypes.c_void_p(salts.ctypes.data) if salts is not None else None, _p(nodes))
    return nodes


# ---- proof of work (pow.tcc) ----------------------------------------------------------------------------------
def pow_bitlen(work_parameter, cost_per_hash):
    f = lib().oracle_pow_bitlen
    f.restype = ctypes.c_size_t
    return int(f(ctypes.c_size_t(work_parameter), ctypes.c_size_t(cost_per_hash)))


def _b32(b):
    b = bytes(b)
    assert len(b) == 32
    return (ctypes.c_uint8 * 32).from_buffer_copy(b)


def pow_verify_blake2b(challenge, pow_, bitlen):
    return bool(lib().oracle_pow_verify_blake2b(_b32(challenge), _b32(pow_), ctypes.c_size_t(bitlen)))


def pow_solve_blake2b(challenge, bitlen):
    """Returns (pow bytes, number of candidates tried)."""
    out = (ctypes.c_uint8 * 32)()
    f = lib().oracle_pow_solve_blake2b
    f.restype = ctypes.c_uint64
    calls = f(_b32(challenge), ctypes.c_size_t(bitlen), out)
    return bytes(out), int(calls)


def pow_verify_poseidon(params, challenge, pow_, bitlen):
    return bool(lib().oracle_pow_verify_poseidon(*params.args(), _p(_c(challenge)), _p(_c(pow_)), ctypes.c_size_t(bitlen)))


def pow_solve_poseidon(params, challenge, bitlen):
    out = np.empty(4, dtype=np.uint64)
    f = lib().oracle_pow_solve_poseidon
    f.restype = ctypes.c_uint64
    calls = f(*params.args(), _p(_c(challenge)), ctypes.c_size_t(bitlen), _p(out))
    return out, int(calls)


# ---- LDT reducer (ldt_reducer_aux.tcc, exponentiation.tcc) -------------------------------------------------------
def subspace_element_powers(basis, shift, exponent):
    basis, shift = _c(basis), _c(shift)
    m, w = basis.shape
    out = np.empty((1 << m, w), dtype=np.uint64)
    lib().oracle_subspace_element_powers(w, _p(basis), ctypes.c_size_t(m), _p(shift), ctypes.c_uint64(exponent), _p(out))
    return out


def _size_array(v):
    return (ctypes.c_size_t * len(v))(*[int(x) for x in v])


def ldt_combine_additive(evals, degrees, coefficients, basis, shift):
    """combined_LDT_virtual_oracle::evaluated_contents over an affine subspace; coefficients = set_random_coefficients' argument."""
    evals = [_c(e) for e in evals]
    coefficients, basis, shift = _c(coefficients), _c(basis), _c(shift)
    m, w = basis.shape
    out = np.empty((1 << m, w), dtype=np.uint64)
    ptrs = (ctypes.c_void_p * len(evals))(*[e.ctypes.data for e in evals])
    lib().oracle_ldt_combine_additive(w, ptrs, ctypes.c_size_t(len(evals)), _size_array(degrees), _p(coefficients), _p(basis),
                                      ctypes.c_size_t(m), _p(shift), _p(out))
    return out


def ldt_combine_fp(evals, degrees, coefficients, order, shift):
    evals = [_c(e) for e in evals]
    coefficients, shift = _c(coefficients), _c(shift)
    out = np.empty((order, 3), dtype=np.uint64)
    ptrs = (ctypes.c_void_p * len(evals))(*[e.ctypes.data for e in evals])
    lib().oracle_ldt_combine_fp(ptrs, ctypes.c_size_t(len(evals)), _size_array(degrees), _p(coefficients), ctypes.c_size_t(order), _p(shift), _p(out))
    return out


def fp_coset_element_powers(order, shift, exponent):
    out = np.empty((order, 3), dtype=np.uint64)
    lib().oracle_fp_coset_element_powers(ctypes.c_size_t(order), _p(_c(shift)), ctypes.c_uint64(exponent), _p(out))
    return out


# ---- Merkle set-membership proofs (merkle_tree.tcc:242-515) -----------------------------------------------------------
def membership_proof_indices(num_leaves, positions):
    """Heap indices of the auxiliary hashes of get_set_membership_proof, in emission order."""
    f = lib().oracle_membership_proof_indices
    f.restype = ctypes.c_size_t
    cap = max(1, len(positions) * max(1, int(num_leaves).bit_length()))
    out = (ctypes.c_size_t * cap)()
    cnt = f(ctypes.c_size_t(num_leaves), _size_array(positions), ctypes.c_size_t(len(positions)), out, ctypes.c_size_t(cap))
    if cnt == ctypes.c_size_t(-1).value:
        raise ValueError("All positions must be between 0 and num_leaves-1.")
    return [int(out[i]) for i in range(cnt)]


def membership_proof_validate(root, num_leaves, positions, leaf_hashes, aux):
    """validate_set_membership_proof for a non-zk BLAKE2b tree; positions sorted unique, leaf_hashes (count, 32) uint8."""
    lh = np.ascontiguousarray(leaf_hashes, dtype=np.uint8).reshape(-1, 32)
    ax = np.ascontiguousarray(aux, dtype=np.uint8).reshape(-1, 32)
    rc = lib().oracle_membership_proof_validate(_b32(root), ctypes.c_size_t(num_leaves), _size_array(positions), ctypes.c_size_t(len(positions)),
                                                ctypes.c_void_p(lh.ctypes.data), ctypes.c_void_p(ax.ctypes.data), ctypes.c_size_t(ax.shape[0]))
    if rc < 0:
        raise AssertionError("Validation did not consume the entire proof.")
    return bool(rc)


def count_hashes_to_verify(num_leaves, positions):
    f = lib().oracle_count_hashes_to_verify
    f.restype = ctypes.c_size_t
    return int(f(ctypes.c_size_t(num_leaves), _size_array(positions), ctypes.c_size_t(len(positions))))


# ---- FRI verifier pieces (fri_aux.tcc:270-303) ------------------------------------------------------------------------
def fri_fold_at_coset(coset_evals, coset_basis, shift, x_i):
    coset_evals, coset_basis, shift, x_i = _c(coset_evals), _c(coset_basis), _c(shift), _c(x_i)
    w = coset_evals.shape[1]
    out = np.empty(w, dtype=np.uint64)
    lib().oracle_fri_fold_at_coset(w, _p(coset_evals), ctypes.c_size_t(coset_evals.shape[0]), _p(coset_basis), ctypes.c_size_t(coset_basis.shape[0]),
                                   _p(shift), _p(x_i), _p(out))
    return out


def poly_eval(coeffs, x):
    coeffs, x = _c(coeffs), _c(x)
    w = coeffs.shape[1]
    out = np.empty(w, dtype=np.uint64)
    lib().oracle_poly_eval(w, _p(coeffs), ctypes.c_size_t(coeffs.shape[0]), _p(x), _p(out))
    return out


def fp_fri_fold_at_coset(coset_evals, g, h, x_i):
    coset_evals = _c(coset_evals)
    out = np.empty(3, dtype=np.uint64)
    lib().oracle_fp_fri_fold_at_coset(_p(coset_evals), ctypes.c_size_t(coset_evals.shape[0]), _p(_c(g)), _p(_c(h)), _p(_c(x_i)), _p(out))
    return out


def fp_poly_eval(coeffs, x):
    coeffs = _c(coeffs)
    out = np.empty(3, dtype=np.uint64)
    lib().oracle_fp_poly_eval(_p(coeffs), ctypes.c_size_t(coeffs.shape[0]), _p(_c(x)), _p(out))
    return out


def fp_pow(a, e):
    out = np.empty(3, dtype=np.uint64)
    lib().oracle_fp_pow(_p(_c(a)), ctypes.c_uint64(e), _p(out))
    return out


# ---- R1CS row check (rowcheck.tcc:16-88) -------------------------------------------------------------------------------
def rowcheck_additive(az, bz, cz, basis, shift, h, constraint_shift):
    az, bz, cz, basis, shift, constraint_shift = _c(az), _c(bz), _c(cz), _c(basis), _c(shift), _c(constraint_shift)
    m, w = basis.shape
    out = np.empty_like(az)
    lib().oracle_rowcheck_additive(w, _p(az), _p(bz), _p(cz), _p(basis), ctypes.c_size_t(m), _p(shift), ctypes.c_size_t(h), _p(constraint_shift), _p(out))
    return out


def rowcheck_fp(az, bz, cz, shift, order_h, constraint_shift):
    az, bz, cz = _c(az), _c(bz), _c(cz)
    out = np.empty_like(az)
    lib().oracle_rowcheck_fp(_p(az), _p(bz), _p(cz), ctypes.c_size_t(az.shape[0]), _p(_c(shift)), ctypes.c_size_t(order_h), _p(_c(constraint_shift)), _p(out))
    return out


def fz_additive(fw, f1v, basis, shift, input_basis, input_shift):
    fw, f1v, basis, shift, ib, ish = _c(fw), _c(f1v), _c(basis), _c(shift), _c(input_basis).reshape(-1, _c(basis).shape[1]), _c(input_shift)
    m, w = basis.shape
    out = np.empty_like(fw)
    lib().oracle_fz_additive(w, _p(fw), _p(f1v), _p(basis), ctypes.c_size_t(m), _p(shift), _p(ib), ctypes.c_size_t(ib.shape[0]), _p(ish), _p(out))
    return out


def fz_fp(fw, f1v, shift, input_order, input_shift):
    fw, f1v = _c(fw), _c(f1v)
    out = np.empty_like(fw)
    lib().oracle_fz_fp(_p(fw), _p(f1v), ctypes.c_size_t(fw.shape[0]), _p(_c(shift)), ctypes.c_size_t(input_order), _p(_c(input_shift)), _p(out))
    return out


def sumcheck_g_additive(f, h, basis, shift, sbasis, sshift, mu):
    f, h, basis, shift, sshift, mu = _c(f), _c(h), _c(basis), _c(shift), _c(sshift), _c(mu)
    sb = _c(sbasis).reshape(-1, basis.shape[1])
    m, w = basis.shape
    out = np.empty_like(f)
    lib().oracle_sumcheck_g_additive(w, _p(f), _p(h), _p(basis), ctypes.c_size_t(m), _p(shift), _p(sb), ctypes.c_size_t(sb.shape[0]), _p(sshift), _p(mu), _p(out))
    return out


def sumcheck_g_fp(f, h, shift, order_h, sshift, mu):
    f, h = _c(f), _c(h)
    out = np.empty_like(f)
    lib().oracle_sumcheck_g_fp(_p(f), _p(h), ctypes.c_size_t(f.shape[0]), _p(_c(shift)), ctypes.c_size_t(order_h), _p(_c(sshift)), _p(_c(mu)), _p(out))
    return out


def lincheck_combine(fz, mz, r, p1, p2, prime_field=False):
    fz, r, p1, p2 = _c(fz), _c(r), _c(p1), _c(p2)
    mz = [_c(v) for v in mz]
    out = np.empty_like(fz)
    ptrs = (ctypes.c_void_p * len(mz))(*[v.ctypes.data for v in mz])
    lib().oracle_lincheck_combine(0 if prime_field else fz.shape[1], _p(fz), ptrs, ctypes.c_size_t(len(mz)), _p(r), _p(p1), _p(p2),
                                  ctypes.c_size_t(fz.shape[0]), _p(out))
    return out


# ---- Aurora SNARK, prover and verifier (oracle/aurora.hpp) ----
FIELD_EDWARDS, FIELD_GF64, FIELD_GF192 = 0, 1, 3
_AURORA_PARAM_NAMES = ("codeword_domain_dim", "pow_bits", "query_soundness_error_bits", "interactive_soundness_error_bits",
                       "max_tested_degree_bound", "max_constraint_degree_bound", "absolute_proximity_parameter", "multi_lincheck_repetitions",
                       "num_output_LDT_instances", "fri_interactive_repetitions", "fri_query_repetitions")


def _aurora_args(field, log_constraints, num_inputs, seed, security, rs_extra, localization):
    sz = ctypes.c_size_t
    return [ctypes.c_int(field), sz(log_constraints), sz(num_inputs), ctypes.c_uint64(seed), sz(security), sz(rs_extra), sz(localization)]


def aurora_prove(field, log_constraints, num_inputs, seed, security=128, rs_extra=5, localization=2):
    """The serialized transcript of aurora_snark_prover on generate_r1cs_example(2^log_constraints, num_inputs, 2^log_constraints - 1; seed)."""
    l = lib()
    l.oracle_aurora_prove.restype = ctypes.c_long
    n = l.oracle_aurora_prove(*_aurora_args(field, log_constraints, num_inputs, seed, security, rs_extra, localization))
    if n < 0:
        raise RuntimeError("oracle_aurora_prove failed (%d)" % n)
    buf = (ctypes.c_uint8 * n)()
    l.oracle_aurora_fetch(buf)
    return bytes(buf)


def aurora_verify(field, log_constraints, num_inputs, seed, transcript, security=128, rs_extra=5, localization=2, primary_override=None):
    """aurora_snark_verifier on the same seeded instance; primary_override replaces the statement's primary input."""
    buf = (ctypes.c_uint8 * len(transcript)).from_buffer_copy(bytes(transcript))
    po = None
    if primary_override is not None:
        po = np.ascontiguousarray(primary_override, dtype=np.uint64)
    rc = lib().oracle_aurora_verify(*_aurora_args(field, log_constraints, num_inputs, seed, security, rs_extra, localization), buf,
                                    ctypes.c_size_t(len(transcript)), _p(po) if po is not None else None)
    if rc < 0:
        raise RuntimeError("oracle_aurora_verify failed (%d)" % rc)
    return bool(rc)


def aurora_params(field, log_constraints, num_inputs, security=128, rs_extra=5, localization=2):
    out = np.zeros(80, dtype=np.uint64)
    sz = ctypes.c_size_t
    n = lib().oracle_aurora_params(ctypes.c_int(field), sz(log_constraints), sz(num_inputs), sz(security), sz(rs_extra), sz(localization), _p(out), sz(80))
    if n < 0:
        raise RuntimeError("oracle_aurora_params failed (%d)" % n)
    d = dict(zip(_AURORA_PARAM_NAMES, (int(v) for v in out[:11])))
    d["localization_parameters"] = [int(v) for v in out[12:n]]
    return d


def r1cs_example(field, log_constraints, num_inputs, seed):
    """(z = primary || auxiliary, C's column index per row, C's coefficient per row) of the seeded instance."""
    words = {FIELD_EDWARDS: 3, FIELD_GF64: 1, FIELD_GF192: 3}[field]
    n = 1 << log_constraints
    z = np.zeros((n - 1, words), dtype=np.uint64)
    idx = np.zeros(n, dtype=np.uint64)
    coeff = np.zeros((n, words), dtype=np.uint64)
    rc = lib().oracle_r1cs_example(ctypes.c_int(field), ctypes.c_size_t(log_constraints), ctypes.c_size_t(num_inputs), ctypes.c_uint64(seed),
                                   _p(z), _p(idx), _p(coeff))
    assert rc == 0
    return z, idx, coeff


def fri_snark_prove(field, codeword_domain_dim, rs_extra, localization, interactions, queries, seed):
    """Serialized transcript of the FRI-only SNARK (oracle/aurora.hpp FRI_snark_prover) on the seeded degree-2^(dim - rs_extra) polynomial."""
    l = lib()
    l.oracle_fri_snark_prove.restype = ctypes.c_long
    sz = ctypes.c_size_t
    n = l.oracle_fri_snark_prove(ctypes.c_int(field), sz(codeword_domain_dim), sz(rs_extra), sz(localization), sz(interactions), sz(queries), ctypes.c_uint64(seed))
    if n < 0:
        raise RuntimeError("oracle_fri_snark_prove failed (%d)" % n)
    buf = (ctypes.c_uint8 * n)()
    l.oracle_aurora_fetch(buf)
    return bytes(buf)


def fri_snark_verify(field, codeword_domain_dim, rs_extra, localization, interactions, queries, transcript):
    sz = ctypes.c_size_t
    buf = (ctypes.c_uint8 * len(transcript)).from_buffer_copy(bytes(transcript))
    rc = lib().oracle_fri_snark_verify(ctypes.c_int(field), sz(codeword_domain_dim), sz(rs_extra), sz(localization), sz(interactions), sz(queries),
                                       buf, sz(len(transcript)))
    if rc < 0:
        raise RuntimeError("oracle_fri_snark_verify failed (%d)" % rc)
    return bool(rc)


# ---- Fractal preprocessing SNARK (oracle/fractal.hpp) ----
_FRACTAL_PARAM_NAMES = ["codeword_domain_dim", "pow_bits", "query_soundness_error_bits", "interactive_soundness_error_bits",
                        "max_LDT_tested_degree_bound", "max_constraint_degree_bound", "absolute_proximity_parameter", "holographic_lincheck_repetitions",
                        "num_output_LDT_instances", "fri_interactive_repetitions", "fri_query_repetitions", "index_domain_dim", "matrix_domain_dim"]


def fractal_prove(field, log_constraints, num_inputs, seed, security=128, rs_extra=3, localization=2):
    """(serialized transcript, index Merkle roots) of fractal_snark_indexer + fractal_snark_prover on the seeded instance
    generate_r1cs_example(2^log_constraints, num_inputs, 2^log_constraints - 1; seed) (profiling/instrument_fractal_snark.cpp:93-160)."""
    l = lib()
    l.oracle_fractal_prove.restype = ctypes.c_long
    l.oracle_fractal_index_roots.restype = ctypes.c_long
    n = l.oracle_fractal_prove(*_aurora_args(field, log_constraints, num_inputs, seed, security, rs_extra, localization))
    if n < 0:
        raise RuntimeError("oracle_fractal_prove failed (%d)" % n)
    buf = (ctypes.c_uint8 * n)()
    l.oracle_aurora_fetch(buf)
    k = l.oracle_fractal_index_roots(None)
    roots = (ctypes.c_uint8 * (32 * k))()
    l.oracle_fractal_index_roots(roots)
    return bytes(buf), [bytes(roots[32 * i:32 * i + 32]) for i in range(k)]


def fractal_verify(field, log_constraints, num_inputs, seed, transcript, index_roots, security=128, rs_extra=3, localization=2, primary_override=None):
    """fractal_snark_verifier with the verifier index (the index trees' roots) on the same seeded instance."""
    buf = (ctypes.c_uint8 * len(transcript)).from_buffer_copy(bytes(transcript))
    flat = b"".join(index_roots)
    rbuf = (ctypes.c_uint8 * max(1, len(flat))).from_buffer_copy(flat if flat else b"\0")
    po = None
    if primary_override is not None:
        po = np.ascontiguousarray(primary_override, dtype=np.uint64)
    rc = lib().oracle_fractal_verify(*_aurora_args(field, log_constraints, num_inputs, seed, security, rs_extra, localization), buf,
                                     ctypes.c_size_t(len(transcript)), rbuf, ctypes.c_size_t(len(index_roots)), _p(po) if po is not None else None)
    if rc < 0:
        raise RuntimeError("oracle_fractal_verify failed (%d)" % rc)
    return bool(rc)


def fractal_params(field, log_constraints, num_inputs, security=128, rs_extra=3, localization=2):
    out = np.zeros(80, dtype=np.uint64)
    sz = ctypes.c_size_t
    n = lib().oracle_fractal_params(ctypes.c_int(field), sz(log_constraints), sz(num_inputs), sz(security), sz(rs_extra), sz(localization), _p(out), sz(80))
    if n < 0:
        raise RuntimeError("oracle_fractal_params failed (%d)" % n)
    d = dict(zip(_FRACTAL_PARAM_NAMES, (int(v) for v in out[:13])))
    d["localization_parameters"] = [int(v) for v in out[14:n]]
    return d


def fractal_index_oracle(field, log_constraints, num_inputs, seed, matrix, which, security=128, rs_extra=3, localization=2):
    """Index oracle `which` (0 row, 1 col, 2 val, 3 row*col) of matrix 0..2 (A, B, C) over the codeword domain."""
    words = {FIELD_EDWARDS: 3, FIELD_GF64: 1, FIELD_GF192: 3}[field]
    dim = fractal_params(field, log_constraints, num_inputs, security, rs_extra, localization)["codeword_domain_dim"]
    out = np.zeros((1 << dim, words), dtype=np.uint64)
    sz = ctypes.c_size_t
    rc = lib().oracle_fractal_index_oracle(*_aurora_args(field, log_constraints, num_inputs, seed, security, rs_extra, localization), sz(matrix), sz(which), _p(out))
    if rc != 0:
        raise RuntimeError("oracle_fractal_index_oracle failed (%d)" % rc)
    return out


# ---- general instances: CSR triples + the full assignment in, transcript out (oracle_capi.cpp "general instances") ----
def _csr_args(field, matrices, num_variables, num_inputs):
    """matrices: three (row_ptr, col, coeff) triples — A, B, C; coeff is (entries, words) uint64.  Returns (ctypes arguments, keep-alive list)."""
    keep, rp, cl, cf = [], (_u64p * 3)(), (ctypes.POINTER(ctypes.c_uint32) * 3)(), (_u64p * 3)()
    words = {FIELD_EDWARDS: 3, FIELD_GF64: 1, FIELD_GF192: 3}[field]
    for q, (row_ptr, col, coeff) in enumerate(matrices):
        a, b = np.ascontiguousarray(row_ptr, dtype=np.uint64), np.ascontiguousarray(col, dtype=np.uint32)
        c = np.ascontiguousarray(coeff, dtype=np.uint64).reshape(-1, words)
        assert c.shape[0] == b.shape[0] == int(a[-1]), "CSR arrays disagree"
        keep += [a, b, c]
        rp[q], cl[q], cf[q] = _p(a), b.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _p(c)
    sz = ctypes.c_size_t
    n = len(matrices[0][0]) - 1
    return [ctypes.c_int(field), sz(n), sz(num_variables), sz(num_inputs), rp, cl, cf], keep


def aurora_prove_csr(field, matrices, num_variables, num_inputs, assignment, security=128, rs_extra=5, localization=2):
    """Serialized transcript of aurora_snark_prover on the caller's constraint system; assignment = (num_variables, words), primary inputs first."""
    l = lib()
    l.oracle_aurora_prove_csr.restype = ctypes.c_long
    args, keep = _csr_args(field, matrices, num_variables, num_inputs)
    z = _c(assignment)
    sz = ctypes.c_size_t
    n = l.oracle_aurora_prove_csr(*args, _p(z), sz(security), sz(rs_extra), sz(localization))
    if n < 0:
        raise RuntimeError("oracle_aurora_prove_csr failed (%d)" % n)
    buf = (ctypes.c_uint8 * n)()
    l.oracle_aurora_fetch(buf)
    return bytes(buf)


def aurora_verify_csr(field, matrices, num_variables, num_inputs, primary_input, transcript, security=128, rs_extra=5, localization=2):
    args, keep = _csr_args(field, matrices, num_variables, num_inputs)
    prim = _c(primary_input) if num_inputs else np.zeros((1, 3), dtype=np.uint64)
    buf = (ctypes.c_uint8 * len(transcript)).from_buffer_copy(bytes(transcript))
    sz = ctypes.c_size_t
    rc = lib().oracle_aurora_verify_csr(*args, _p(prim), sz(security), sz(rs_extra), sz(localization), buf, sz(len(transcript)))
    if rc < 0:
        raise RuntimeError("oracle_aurora_verify_csr failed (%d)" % rc)
    return bool(rc)


def fractal_prove_csr(field, matrices, num_variables, num_inputs, assignment, security=128, rs_extra=3, localization=2):
    """(serialized transcript, index Merkle roots) of fractal_snark_indexer + fractal_snark_prover on the caller's constraint system."""
    l = lib()
    l.oracle_fractal_prove_csr.restype = ctypes.c_long
    l.oracle_fractal_index_roots.restype = ctypes.c_long
    args, keep = _csr_args(field, matrices, num_variables, num_inputs)
    z = _c(assignment)
    sz = ctypes.c_size_t
    n = l.oracle_fractal_prove_csr(*args, _p(z), sz(security), sz(rs_extra), sz(localization))
    if n < 0:
        raise RuntimeError("oracle_fractal_prove_csr failed (%d)" % n)
    buf = (ctypes.c_uint8 * n)()
    l.oracle_aurora_fetch(buf)
    k = l.oracle_fractal_index_roots(None)
    roots = (ctypes.c_uint8 * (32 * k))()
    l.oracle_fractal_index_roots(roots)
    return bytes(buf), [bytes(roots[32 * i:32 * i + 32]) for i in range(k)]


def fractal_verify_csr(field, matrices, num_variables, num_inputs, primary_input, transcript, index_roots, security=128, rs_extra=3, localization=2):
    args, keep = _csr_args(field, matrices, num_variables, num_inputs)
    prim = _c(primary_input) if num_inputs else np.zeros((1, 3), dtype=np.uint64)
    buf = (ctypes.c_uint8 * len(transcript)).from_buffer_copy(bytes(transcript))
    flat = b"".join(index_roots)
    rbuf = (ctypes.c_uint8 * max(1, len(flat))).from_buffer_copy(flat if flat else b"\0")
    sz = ctypes.c_size_t
    rc = lib().oracle_fractal_verify_csr(*args, _p(prim), sz(security), sz(rs_extra), sz(localization), buf, sz(len(transcript)), rbuf, sz(len(index_roots)))
    if rc < 0:
        raise RuntimeError("oracle_fractal_verify_csr failed (%d)" % rc)
    return bool(rc)


def r1cs_check_csr(field, matrices, num_variables, num_inputs, assignment):
    """(number of violated constraints, Az, Bz, Cz) for z = (1, assignment): r1cs_constraint_system::is_satisfied and
    create_Az_Bz_Cz_from_variable_assignment."""
    l = lib()
    l.oracle_r1cs_check_csr.restype = ctypes.c_long
    args, keep = _csr_args(field, matrices, num_variables, num_inputs)
    words = {FIELD_EDWARDS: 3, FIELD_GF64: 1, FIELD_GF192: 3}[field]
    n = len(matrices[0][0]) - 1
    z = _c(assignment)
    out = np.zeros((3, n, words), dtype=np.uint64)
    bad = l.oracle_r1cs_check_csr(*args, _p(z), _p(out))
    if bad < 0:
        raise RuntimeError("oracle_r1cs_check_csr failed (%d)" % bad)
    return int(bad), out[0], out[1], out[2]


def block_times(reset=True):
    """{block name: (inclusive seconds, calls)} of the oracle provers since the last reset, under the reference's libff::enter_block names
    ("Construct Merkle tree", "pow", "evaluating next FRI codeword", "Call to additive_FFT_wrapper", ...: oracle/field.hpp)."""
    l = lib()
    l.oracle_block_times.restype = ctypes.c_size_t
    need = l.oracle_block_times(None, ctypes.c_size_t(0), ctypes.c_int(0))
    buf = ctypes.create_string_buffer(need + 16)
    l.oracle_block_times(buf, ctypes.c_size_t(need + 16), ctypes.c_int(1 if reset else 0))
    out = {}
    for line in buf.value.decode().splitlines():
        name, sec, calls = line.split("\t")
        out[name] = (float(sec), int(calls))
    return out
