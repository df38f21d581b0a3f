"""End-to-end FRI cases shared by the CPU-emulation and GPU suites: the product's prover (libiop_amd/fri.py over the C ABI)
against the independent oracle-based verifier (tests/fri_verifier.py)."""
import copy

import numpy as np

import oracle
from helpers import rand_elems
import fri_verifier


class _HostTensor:
    """Just enough of a torch tensor for libiop_amd.fri on the CPU emulation: a numpy array whose data_ptr() is its address
    (the emulated library's "device" memory is host memory)."""

    def __init__(self, arr):
        self.a = arr
        self.shape = arr.shape
        self.device = None

    def data_ptr(self):
        return self.a.ctypes.data

    def __getitem__(self, k):
        return _HostTensor(self.a[k])

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class _HostTorch:
    uint8, int64 = np.uint8, np.int64

    @staticmethod
    def empty(shape, dtype=None, device=None):
        return _HostTensor(np.zeros(shape, dtype=dtype))

    @staticmethod
    def empty_like(t):
        return _HostTensor(np.zeros_like(t.a))


def prove_and_verify(lib, torch, to_device, m, rs_extra, loc_param, num_queries, pow_bitlen, seed, kind="standard"):
    import libiop_amd.fri as fri
    import libiop_amd.host as host
    d = m - rs_extra
    if kind == "standard":
        basis, shift = oracle.standard_basis(m, 3), np.array([1 << m, 0, 0], dtype=np.uint64)
    else:
        basis, shift = rand_elems(seed + 1, m, 3), rand_elems(seed + 2, 1, 3)[0]
    coeffs = rand_elems(seed, 1 << d, 3)
    codeword = lib.additive_FFT(coeffs, basis, shift)               # a codeword of degree < 2^d
    loc = host.localization_parameter_to_array(loc_param, m, rs_extra)
    final_bound = max(1, (1 << d) >> sum(loc))          # degree bound left after all reductions (fri_ldt.tcc:534-543)
    proof = fri.fri_prove(lib, torch, to_device(codeword), basis, shift, loc, final_bound, num_queries, pow_bitlen)
    ok, why = fri_verifier.verify(proof, basis, shift, loc, final_bound, num_queries, pow_bitlen)
    assert ok, why
    # soundness smoke: every tampered component is rejected (test_fri.cpp's invalid-proof cases)
    for field in ["roots", "final_polynomial", "proof_of_work", "query_responses", "membership_proofs"]:
        bad = copy.deepcopy(proof)
        v = getattr(bad, field)
        if field == "roots":
            v[len(v) // 2] = bytes(32)
        elif field == "proof_of_work":
            bad.proof_of_work = bytes(32) if pow_bitlen > 8 else None
            if bad.proof_of_work is None:
                continue
        elif field == "final_polynomial":
            v[0, 0] ^= np.uint64(1)
        elif field == "query_responses":
            v[-1][0, 0, 0] ^= np.uint64(1)
        else:
            nonempty = [k for k in range(len(v)) if len(v[k])]
            if not nonempty:
                continue
            v[nonempty[0]][0, 0] ^= 1
        ok, why = fri_verifier.verify(bad, basis, shift, loc, final_bound, num_queries, pow_bitlen)
        assert not ok, field
    # a codeword that is far from low degree: the honest prover's transcript is rejected
    far = rand_elems(seed + 9, 1 << m, 3)
    proof = fri.fri_prove(lib, torch, to_device(far), basis, shift, loc, final_bound, num_queries, pow_bitlen)
    ok, why = fri_verifier.verify(proof, basis, shift, loc, final_bound, num_queries, pow_bitlen)
    assert not ok
    return True


def host_env():
    return _HostTorch, (lambda arr: _HostTensor(np.ascontiguousarray(arr).view(np.int64)))


def prove_and_verify_multiplicative(lib, torch, to_device, log_n, rs_extra, loc_param, num_queries, pow_bitlen, seed):
    import libiop_amd as la
    import libiop_amd.fri as fri
    import libiop_amd.host as host
    d = log_n - rs_extra
    P = la.EDWARDS_FR_MODULUS
    rng = np.random.default_rng(seed)
    coeffs = la.edwards_to_montgomery([int.from_bytes(rng.bytes(32), "little") % P for _ in range(1 << d)])
    shift_int = la.EDWARDS_FR_GENERATOR
    codeword = lib.multiplicative_FFT(coeffs, log_n, la.edwards_to_montgomery([shift_int])[0])
    loc = host.localization_parameter_to_array(loc_param, log_n, rs_extra)
    final_bound = max(1, (1 << d) >> sum(loc))
    args = (log_n, shift_int, loc, final_bound, num_queries, pow_bitlen)
    proof = fri.fri_prove_multiplicative(lib, torch, to_device(codeword), *args)
    ok, why = fri_verifier.verify_multiplicative(proof, *args)
    assert ok, why
    for field in ["roots", "final_polynomial", "query_responses", "membership_proofs"]:
        bad = copy.deepcopy(proof)
        v = getattr(bad, field)
        if field == "roots":
            v[0] = bytes(32)
        elif field == "final_polynomial":
            v[0, 0] ^= np.uint64(1)
        elif field == "query_responses":
            v[-1][0, 0, 0] ^= np.uint64(1)
        else:
            nonempty = [k for k in range(len(v)) if len(v[k])]
            if not nonempty:
                continue
            v[nonempty[0]][0, 0] ^= 1
        ok, why = fri_verifier.verify_multiplicative(bad, *args)
        assert not ok, field
    far = la.edwards_to_montgomery([int.from_bytes(rng.bytes(32), "little") % P for _ in range(1 << log_n)])
    proof = fri.fri_prove_multiplicative(lib, torch, to_device(far), *args)
    ok, why = fri_verifier.verify_multiplicative(proof, *args)
    assert not ok
    return True
