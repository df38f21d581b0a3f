"""Host-side (log-size) pieces of the accelerated path, in Python: GF(2^192) arithmetic on ints for domain
metadata, the FRI domain chain and localization array, and the BLAKE2b hashchain that produces the verifier
challenges.  None of this touches codeword-sized data; BLAKE2b comes from hashlib (RFC 7693, what libsodium's
crypto_generichash_blake2b computes).  Citations are relative to the reference tree."""
import hashlib

import numpy as np

GF192_TAIL = 0x87                      # x^192 = x^7 + x^2 + x + 1 (libff gf192)
_MASK192 = (1 << 192) - 1


def gf_from_words(w):
    return int(w[0]) | (int(w[1]) << 64) | (int(w[2]) << 128)


def gf_to_words(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(3)], dtype=np.uint64)


def gf_mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    # reduce modulo x^192 + x^7 + x^2 + x + 1
    while r >> 192:
        hi = r >> 192
        r = (r & _MASK192) ^ hi ^ (hi << 1) ^ (hi << 2) ^ (hi << 7)
    return r


def gf_sq(a):
    return gf_mul(a, a)


def gf_inv(a):
    r = a                               # a^(2^192 - 2)
    for _ in range(190):
        r = gf_mul(gf_sq(r), a)
    return gf_sq(r)


def localization_parameter_to_array(localization_parameter, codeword_domain_dim, rs_extra_dimensions):
    """FRI_protocol_parameters::localization_parameter_to_array — fri_ldt.tcc:132-146."""
    num_reductions = ((codeword_domain_dim - rs_extra_dimensions - 1) // localization_parameter) + 1
    return [1] + [localization_parameter] * (num_reductions - 1)


def _vanishing_poly_of_span(basis):
    """vanishing_polynomial_from_subspace for an unshifted subspace (vanishing_polynomial.tcc:373-395):
    coefficient i >= 1 multiplies X^(2^(i-1)), slot 0 is the constant term."""
    poly = [0, 1]
    for c in basis:
        pc = _lin_eval(poly, c)
        sq = [0] * (len(poly) + 1)
        for i in range(len(poly), 0, -1):
            sq[i] = gf_sq(poly[i - 1])
        for i in range(len(poly)):
            sq[i] ^= gf_mul(poly[i], pc)
        poly = sq
    return poly


def _lin_eval(poly, x):
    r = poly[0]
    xp = x
    for c in poly[1:]:
        r ^= gf_mul(c, xp)
        xp = gf_sq(xp)
    return r


def fri_additive_domains(basis, shift, localization_parameters):
    """FRI_protocol::compute_domains, additive branch — fri_ldt.tcc:310-338.  basis: (m, 3) words, shift: (3,).
    Returns [(basis_i, shift_i)] for L^(0) (the codeword domain), L^(1), ..."""
    b = [gf_from_words(w) for w in np.asarray(basis, dtype=np.uint64)]
    s = gf_from_words(np.asarray(shift, dtype=np.uint64))
    out = [(np.array([gf_to_words(v) for v in b], dtype=np.uint64).reshape(-1, 3), gf_to_words(s))]
    for eta in localization_parameters:
        q = _vanishing_poly_of_span(b[:eta])
        s = _lin_eval(q, s)
        b = [_lin_eval(q, v) for v in b[eta:]]
        out.append((np.array([gf_to_words(v) for v in b], dtype=np.uint64).reshape(-1, 3), gf_to_words(s)))
    return out


class Blake2bHashchain:
    """blake2b_hashchain (libiop/bcs/hashing/blake2b.tcc:10-110), 32-byte state, including the reference's
    behaviour that absorb() hashes only the first digest_len bytes of state || input (:56-60): the state
    advances to BLAKE2b-256(state) whatever is absorbed — reproduced, not fixed."""

    DIGEST_LEN = 32

    def __init__(self):
        self.state = b" " * self.DIGEST_LEN                  # :17
        self.squeeze_index = 0

    def absorb(self, _data=None):
        buf = self.state + (bytes(_data) if _data is not None else b"")
        self.state = hashlib.blake2b(buf[: self.DIGEST_LEN], digest_size=self.DIGEST_LEN).digest()

    def squeeze_gf192(self, num_elements):
        """:76-86, :162-185, :231-257 — element i = keyed BLAKE2b(state || index, key = i, 24 bytes), raw words."""
        self.squeeze_index += 1
        msg = self.state + self.squeeze_index.to_bytes(8, "little")
        out = np.zeros((num_elements, 3), dtype=np.uint64)
        for i in range(num_elements):
            d = hashlib.blake2b(msg, digest_size=24, key=i.to_bytes(8, "little")).digest()
            out[i] = np.frombuffer(d, dtype=np.uint64)
        return out

    def squeeze_root_type(self):
        """:105-110 — one squeezed element hashed to a digest (blake2b_field_element_hash, :140-160)."""
        x = self.squeeze_gf192(1)
        return hashlib.blake2b(x.tobytes(), digest_size=self.DIGEST_LEN).digest()

    def squeeze_query_positions(self, num_positions, range_of_positions):
        """:88-105 + blake2b.cpp:50-74."""
        if range_of_positions & (range_of_positions - 1):
            raise ValueError("upper_bound must be a power of two.")
        out = []
        for _ in range(num_positions):
            self.squeeze_index += 1
            d = hashlib.blake2b(self.state, digest_size=8, key=self.squeeze_index.to_bytes(8, "little")).digest()
            out.append(int.from_bytes(d, "little") % range_of_positions)
        return out
