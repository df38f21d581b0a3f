"""Evaluation domains and the two field arms of the accelerated path, host side (O(log n) metadata only).

`Domain` mirrors the reference's tagged union field_subset<FieldT> (libiop/algebra/field_subset/field_subset.tcc:3-62,
130-237): an affine subspace of GF(2^192) (subspace.tcc) or a multiplicative coset of the 181-bit prime field
(subgroup.tcc).  `GF192` / `EdwardsFr` carry what differs between the arms — scalar arithmetic on Python ints for the
handful of host-side constants, the hashchain extractor (blake2b.tcc:162-257), and the dispatch of every device operator
to the matching C-ABI entry point (FFT_over_field_subset and friends dispatch on the domain type the same way,
fft.tcc:407-475, fri_aux.tcc:5-34).  Citations are relative to the reference tree."""
import hashlib

import numpy as np

import libiop_amd as la
from . import host

ADDITIVE, MULTIPLICATIVE = "affine_subspace", "multiplicative_coset"


def _log2(n):
    return max(int(n) - 1, 0).bit_length()          # libff::log2 is the ceiling log


class Domain:
    """field_subset<FieldT>."""

    def __init__(self, field, kind, basis=None, shift=None, log_n=None):
        self.field, self.kind = field, kind
        if kind == ADDITIVE:
            self.basis = np.ascontiguousarray(basis, dtype=np.uint64).reshape(-1, 3)
            self.shift = np.ascontiguousarray(shift, dtype=np.uint64).reshape(3)
            self.dim = self.basis.shape[0]
        else:
            self.dim = int(log_n)
            self.shift_int = int(shift) % la.EDWARDS_FR_MODULUS          # canonical integer
            if self.shift_int == 0:
                raise ValueError("coset_shift was supplied as 0, it was likely intended to be 1")
            self.shift = la.edwards_to_montgomery([self.shift_int])[0]
            self.gen = la.edwards_subgroup_generator(self.dim)          # subgroup.tcc:55-59
        self.size = 1 << self.dim

    # ---- field_subset.tcc ----
    @property
    def additive(self):
        return self.kind == ADDITIVE

    @property
    def domain_type(self):
        return la.DOMAIN_ADDITIVE if self.additive else la.DOMAIN_MULTIPLICATIVE

    def num_elements(self):
        return self.size

    def dimension(self):
        return self.dim

    def get_subset_of_order(self, order):
        """:217-237 — first log2(order) basis vectors, same shift / the default subgroup of that order, same shift."""
        d = _log2(order)
        if self.additive:
            return Domain(self.field, ADDITIVE, basis=self.basis[:d], shift=self.shift)
        return Domain(self.field, MULTIPLICATIVE, shift=self.shift_int, log_n=d)

    def element_outside_of_subset(self):
        """subspace.tcc:219-227 (standard basis): shift + FieldT(1 << dim); subgroup.tcc:311-315: shift * multiplicative_generator."""
        if self.additive:
            if not np.array_equal(self.basis, la.standard_basis(self.dim)):
                raise ValueError("subspace.element_outside_of_subset() is only supported for standard basis")
            return host.gf_to_words(host.gf_from_words(self.shift) ^ (1 << self.dim))
        return (self.shift_int * la.EDWARDS_FR_GENERATOR) % la.EDWARDS_FR_MODULUS

    def reindex_by_subset(self, reindex_subset_dim, index):
        """:130-142; subgroup.tcc:149-173."""
        if self.additive:
            return index
        order_s, g_over_s = 1 << reindex_subset_dim, 1 << (self.dim - reindex_subset_dim)
        if index < order_s:
            return index * g_over_s
        i = index - order_s
        return i + (i // (g_over_s - 1)) + 1

    def reindex_by_subset_array(self, reindex_subset_dim, count):
        """reindex_by_subset(dim, i) for i < count, vectorised."""
        idx = np.arange(count, dtype=np.int64)
        if self.additive:
            return idx
        order_s, g_over_s = 1 << reindex_subset_dim, 1 << (self.dim - reindex_subset_dim)
        i = idx - order_s
        return np.where(idx < order_s, idx * g_over_s, i + (i // max(g_over_s - 1, 1)) + 1)

    # coset index maps (subspace.tcc:73-91, subgroup.tcc:175-197)
    def coset_index(self, position, coset_size):
        return position // coset_size if self.additive else position % (self.size // coset_size)

    def intra_coset_index(self, position, coset_size):
        return position % coset_size if self.additive else position // (self.size // coset_size)

    def position_by_coset_indices(self, coset_index, intra_coset_index, coset_size):
        if self.additive:
            return coset_index * coset_size + intra_coset_index
        return coset_index + intra_coset_index * (self.size // coset_size)


class GF192:
    """libff::gf192, additive arm."""
    name, additive, elem_bytes, soundness_bits = "gf192", True, 24, 192

    def domain(self, num_elements, shift=None):
        """field_subset(num_elements[, shift]) — field_subset.tcc:3-18,45-62."""
        d = _log2(num_elements)
        return Domain(self, ADDITIVE, basis=la.standard_basis(d), shift=np.zeros(3, dtype=np.uint64) if shift is None else shift)

    # host scalars: (3,) uint64 words
    def zero(self):
        return np.zeros(3, dtype=np.uint64)

    def mul(self, a, b):
        return host.gf_to_words(host.gf_mul(host.gf_from_words(a), host.gf_from_words(b)))

    def add(self, a, b):
        return np.bitwise_xor(np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64))

    def sub(self, a, b):
        return self.add(a, b)

    def neg(self, a):
        return np.asarray(a, dtype=np.uint64)

    def one(self):
        return np.array([1, 0, 0], dtype=np.uint64)

    def inv(self, a, lib):
        return lib.gf192_inverse_host(a)

    def vanishing_eval(self, domain, x, lib):
        """Z_S(x) for the affine subspace S (vanishing_polynomial::evaluation_at_point)."""
        return lib.gf192_vanishing_host(domain.basis, domain.shift, x)[0]

    def vanishing_derivative(self, domain, x, lib):
        """(DZ_S)(x): the linear coefficient (vanishing_polynomial.tcc:63-72)."""
        return lib.gf192_vanishing_host(domain.basis, domain.shift, x)[1]

    def element_in_domain(self, domain, x):
        v = host.gf_from_words(self.add(x, domain.shift))
        if not np.array_equal(domain.basis, la.standard_basis(domain.dim)):
            raise NotImplementedError("membership test for a non-standard basis")
        return v < (1 << domain.dim)

    def squeeze(self, hashchain, n):
        return hashchain.squeeze_gf192(n)

    def fri_domains(self, domain, localization, lib):
        """FRI_protocol::compute_domains, additive branch (fri_ldt.tcc:310-338), by the library's host-side helper."""
        chain = lib.fri_additive_domains(domain.basis, domain.shift, localization)
        return [domain] + [Domain(self, ADDITIVE, basis=b, shift=s) for b, s in chain[1:]]


class EdwardsFr:
    """libff::edwards_Fr (181 bits, 3 Montgomery limbs), multiplicative arm."""
    name, additive, elem_bytes, soundness_bits = "edwards_Fr", False, 24, 180
    P = la.EDWARDS_FR_MODULUS

    def domain(self, num_elements, shift=None):
        return Domain(self, MULTIPLICATIVE, shift=1 if shift is None else shift, log_n=_log2(num_elements))

    def to_int(self, words):
        v = int(words[0]) | (int(words[1]) << 64) | (int(words[2]) << 128)
        return v * pow(1 << 192, -1, self.P) % self.P

    def from_int(self, v):
        return la.edwards_to_montgomery([int(v) % self.P])[0]

    def zero(self):
        return np.zeros(3, dtype=np.uint64)

    def mul(self, a, b):
        return self.from_int(self.to_int(a) * self.to_int(b))

    def add(self, a, b):
        return self.from_int(self.to_int(a) + self.to_int(b))

    def sub(self, a, b):
        return self.from_int(self.to_int(a) - self.to_int(b))

    def neg(self, a):
        return self.from_int(-self.to_int(a))

    def one(self):
        return self.from_int(1)

    def inv(self, a, lib=None):
        return self.from_int(pow(self.to_int(a), -1, self.P))

    def vanishing_eval(self, domain, x, lib=None):
        """Z_S(x) = x^|S| - shift^|S| (vanishing_polynomial.tcc:14-25)."""
        return self.from_int(pow(self.to_int(x), domain.size, self.P) - pow(domain.shift_int, domain.size, self.P))

    def vanishing_derivative(self, domain, x, lib=None):
        """|S| x^(|S| - 1) (vanishing_polynomial.tcc:57-62)."""
        return self.from_int(domain.size * pow(self.to_int(x), domain.size - 1, self.P))

    def element_in_domain(self, domain, x):
        return pow(self.to_int(x) * pow(domain.shift_int, -1, self.P) % self.P, domain.size, self.P) == 1

    def squeeze(self, hashchain, n):
        """blake2b_FieldT_randomness_extractor for Fp (blake2b.tcc:187-257): keyed BLAKE2b straight into mont_repr, bits above
        the modulus MSB cleared, retry with key += num_elements until below p."""
        hashchain.squeeze_index += 1
        msg = hashchain.state + hashchain.squeeze_index.to_bytes(8, "little")
        out = np.zeros((n, 3), dtype=np.uint64)
        mask = (1 << self.P.bit_length()) - 1
        for i in range(n):
            key = i
            while True:
                raw = int.from_bytes(hashlib.blake2b(msg, digest_size=24, key=key.to_bytes(8, "little")).digest(), "little") & mask
                key += n
                if raw < self.P:
                    break
            out[i] = [(raw >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)]
        return out

    def fri_domains(self, domain, localization, lib=None):
        """fri_ldt.tcc:292-308: size >>= eta, shift <- shift^(2^eta)."""
        out, sh, logn = [domain], domain.shift_int, domain.dim
        for eta in localization:
            sh, logn = pow(sh, 1 << eta, self.P), logn - eta
            out.append(Domain(self, MULTIPLICATIVE, shift=sh, log_n=logn))
        return out


class MerkleTree:
    """A BCS Merkle tree resident on the device: (2L - 1, 32) uint8 nodes in heap order (merkle_tree.tcc:92-229)."""

    def __init__(self, lib, nodes, num_leaves):
        self.lib, self.nodes, self.num_leaves = lib, nodes, num_leaves

    def root(self):
        """merkle_tree::get_root, read on the library's stream."""
        return self.lib.read_digest(self.nodes.data_ptr())

    def membership_proof(self, leaf_positions):
        """merkle_tree::get_set_membership_proof (merkle_tree.tcc:242-336): the auxiliary hashes as a (count, 32) uint8 array."""
        return self.lib.get_set_membership_proof_dev(self.nodes.data_ptr(), self.num_leaves, leaf_positions)


class DeviceOps:
    """The device operators of the path, dispatched on the domain type.  Vectors are (count, 3) int64 torch tensors on the
    device the library is bound to; every call enqueues on the library's stream and returns without synchronising."""

    def __init__(self, lib, torch, device, field):
        self.lib, self.torch, self.device, self.field = lib, torch, device, field
        # The provers interleave torch ops (copies, index assignments, torch.cat, buffers recycled by the caching allocator) with
        # library kernels WITHOUT host synchronisation: that is only ordered when both enqueue on the same stream.  Refuse the
        # unshared configuration instead of racing (libiop_amd/dist.py's collectives have their own host-synchronised fallback).
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        if dev.type == "cuda" and not lib.shares_stream_with(torch, dev):
            raise RuntimeError("DeviceOps: the library must enqueue on torch's current stream — call "
                               "lib.set_stream(torch.cuda.current_stream().cuda_stream) before constructing the operators")

    # ---- layout: one GPU holds every vector whole (libiop_amd/dist.py overrides these for contiguous-coset sharding) ----
    def local_size(self, domain):
        return domain.size

    def mark_codeword_domain(self, domain):
        return domain

    def mark_fri_domains(self, domains, localization):
        return domains

    def query_responses(self, d_oracles, domain, positions):
        """values[p][k] = oracle_k[positions[p]] (bcs_prover.tcc:187-197) as a (positions, oracles, 3) uint64 array."""
        return self.lib.query_responses_dev([t.data_ptr() for t in d_oracles], 24, domain.size, positions)

    def empty(self, n):
        return self.torch.empty((max(int(n), 1), 3), dtype=self.torch.int64, device=self.device)[: int(n)]

    def upload(self, host_words):
        a = np.ascontiguousarray(host_words, dtype=np.uint64).reshape(-1, 3)
        t = self.empty(a.shape[0])
        if a.shape[0]:
            self.lib.h2d(t.data_ptr(), a)
        return t

    def upload_raw(self, arr, dtype):
        a = np.ascontiguousarray(arr)
        t = self.torch.empty((max(a.size, 1),), dtype=dtype, device=self.device)
        if a.size:
            self.lib.h2d(t.data_ptr(), a)
        return t

    def download(self, t, count=None):
        n = t.shape[0] if count is None else count
        out = np.empty((n, 3), dtype=np.uint64)
        if n:
            self.lib.d2h(out, t.data_ptr())
        return out

    # ---- transforms ----
    def FFT(self, d_coeffs, n_coeffs, domain):
        """FFT_over_field_subset (fft.tcc:407-419)."""
        out = self.empty(domain.size)
        if domain.additive:
            self.lib.additive_FFT_dev(d_coeffs.data_ptr(), int(n_coeffs), domain.basis, domain.shift, out.data_ptr())
        else:
            self.lib._check(self.lib.c.iopx_mul_fft_fp3_dev(d_coeffs.data_ptr(), int(n_coeffs), domain.dim, _p(domain.gen), _p(domain.shift), out.data_ptr()))
        return out

    def FFT_batch(self, d_coeffs_list, n_coeffs, domain):
        """Several polynomials with the same coefficient count onto one domain."""
        if domain.additive and len(d_coeffs_list) > 1:
            outs = [self.empty(domain.size) for _ in d_coeffs_list]
            d = max(int(n_coeffs) - 1, 0).bit_length()
            self.lib.additive_LDE_batch_dev([t.data_ptr() for t in d_coeffs_list], int(n_coeffs), domain.basis, domain.shift, 0,
                                            1 << (domain.dim - d), [o.data_ptr() for o in outs])
            return outs
        return [self.FFT(t, n_coeffs, domain) for t in d_coeffs_list]

    def IFFT(self, d_evals, domain):
        """IFFT_over_field_subset (fft.tcc:421-433)."""
        out = self.empty(domain.size)
        if domain.additive:
            self.lib.additive_IFFT_dev(d_evals.data_ptr(), domain.basis, domain.shift, out.data_ptr())
        else:
            self.lib._check(self.lib.c.iopx_mul_ifft_fp3_dev(d_evals.data_ptr(), domain.dim, _p(domain.gen), _p(domain.shift), out.data_ptr()))
        return out

    def IFFT_batch(self, d_evals_list, domain):
        if domain.additive and len(d_evals_list) > 1:
            packed = self.empty(domain.size * len(d_evals_list))
            for k, t in enumerate(d_evals_list):
                packed[k * domain.size:(k + 1) * domain.size].copy_(t)
            out = self.empty(domain.size * len(d_evals_list))
            self.lib.additive_IFFT_batch_dev(packed.data_ptr(), len(d_evals_list), domain.basis, domain.shift, out.data_ptr())
            return [out[k * domain.size:(k + 1) * domain.size] for k in range(len(d_evals_list))]
        return [self.IFFT(t, domain) for t in d_evals_list]

    def IFFT_batch_packed(self, packed, batch, domain):
        """`batch` vectors stored back to back in one tensor: one batched inverse transform (phase-1 passes shared)."""
        n = domain.size
        if domain.additive and batch > 1:
            out = self.empty(n * batch)
            self.lib.additive_IFFT_batch_dev(packed.data_ptr(), batch, domain.basis, domain.shift, out.data_ptr())
            return [out[k * n:(k + 1) * n] for k in range(batch)]
        return [self.IFFT(packed[k * n:(k + 1) * n], domain) for k in range(batch)]

    def reextend_packed(self, packed, batch, eval_domain, codeword_domain):
        """FFT_over_field_subset(IFFT_over_field_subset(v, eval_domain), codeword_domain) for `batch` vectors stored back to back.
        When the evaluation domain is spanned by the first basis vectors of the codeword domain the coefficient form is skipped
        (iopx_add_reextend_gf192_batch_dev); otherwise the two transforms run one after the other."""
        n = eval_domain.size
        if codeword_domain.additive and np.array_equal(eval_domain.basis, codeword_domain.basis[: eval_domain.dim]):
            lo, cnt = self._coset_range(codeword_domain, eval_domain.dim)
            outs = [self.empty(cnt << eval_domain.dim) for _ in range(batch)]
            self.lib.additive_reextend_batch_dev(packed.data_ptr(), batch, codeword_domain.basis, eval_domain.dim, eval_domain.shift,
                                                 codeword_domain.shift, lo, cnt, [o.data_ptr() for o in outs])
            return outs
        return self.FFT_batch(self.IFFT_batch_packed(packed, batch, eval_domain), n, codeword_domain)

    def _coset_range(self, codeword_domain, d):
        """The cosets of span(basis[:d]) this process holds: all of them on one GPU."""
        return 0, 1 << (codeword_domain.dim - d)

    def IFFT_of_known_degree(self, d_evals, degree, domain):
        """IFFT_of_known_degree_over_field_subset (fft.tcc:435-475): 2^ceil(log2 degree) coefficients."""
        k = _log2(degree)
        if domain.additive:
            return self.IFFT(d_evals[: 1 << k], domain.get_subset_of_order(1 << k))
        out = self.empty(1 << k)
        self.lib._check(self.lib.c.iopx_mul_ifft_known_degree_fp3_dev(d_evals.data_ptr(), int(degree), domain.dim, _p(domain.gen), _p(domain.shift),
                                                                      out.data_ptr()))
        return out

    # ---- FRI / Merkle ----
    def fold(self, d_f, domain, coset_size, x_i, next_domain=None):
        """evaluate_next_f_i_over_entire_domain (fri_aux.tcc:5-34)."""
        out = self.empty(domain.size // coset_size)
        if domain.additive:
            self.lib.fri_fold_dev(d_f.data_ptr(), domain.basis, domain.shift, coset_size, x_i, out.data_ptr())
        else:
            self.lib._check(self.lib.c.iopx_fri_fold_mul_fp3_dev(d_f.data_ptr(), domain.dim, _p(domain.gen), _p(domain.shift), int(coset_size),
                                                                 _p(x_i), out.data_ptr()))
        return out

    def solve_pow(self, challenge, pow_bitlen):
        """pow::solve_pow (bcs/pow.tcc:67-103): the first passing candidate in the reference's order."""
        return self.lib.solve_pow(challenge, pow_bitlen)

    def merkle_tree(self, d_oracles, domain, coset_size):
        """construct_with_leaves_serialized_by_cosets + compute_inner_nodes over device-resident oracles."""
        leaves = domain.size // coset_size
        nodes = self.torch.empty((2 * leaves - 1, 32), dtype=self.torch.uint8, device=self.device)
        self.lib.merkle_tree_dev([t.data_ptr() for t in d_oracles], 24, domain.size, coset_size, nodes.data_ptr(), domain_type=domain.domain_type)
        return MerkleTree(self.lib, nodes, leaves)

    # ---- virtual oracles ----
    def rowcheck(self, d_az, d_bz, d_cz, codeword_domain, constraint_domain):
        out = self.empty(codeword_domain.size)
        if codeword_domain.additive:
            self.lib.rowcheck_dev(d_az.data_ptr(), d_bz.data_ptr(), d_cz.data_ptr(), codeword_domain.basis, codeword_domain.shift,
                                  constraint_domain.dim, constraint_domain.shift, out.data_ptr())
        else:
            self.lib.rowcheck_multiplicative_dev(d_az.data_ptr(), d_bz.data_ptr(), d_cz.data_ptr(), codeword_domain.dim, codeword_domain.gen,
                                                 codeword_domain.shift, constraint_domain.dim, constraint_domain.shift, out.data_ptr())
        return out

    def fz(self, d_fw, d_f1v, codeword_domain, input_domain):
        out = self.empty(codeword_domain.size)
        if codeword_domain.additive:
            self.lib.fz_dev(d_fw.data_ptr(), d_f1v.data_ptr(), codeword_domain.basis, codeword_domain.shift, input_domain.basis, input_domain.shift,
                            out.data_ptr())
        else:
            self.lib.fz_multiplicative_dev(d_fw.data_ptr(), d_f1v.data_ptr(), codeword_domain.dim, codeword_domain.gen, codeword_domain.shift,
                                           input_domain.dim, input_domain.shift, out.data_ptr())
        return out

    def sumcheck_g(self, d_f, d_h, codeword_domain, summation_domain, claimed_sum):
        out = self.empty(codeword_domain.size)
        if codeword_domain.additive:
            self.lib.sumcheck_g_dev(d_f.data_ptr(), d_h.data_ptr(), codeword_domain.basis, codeword_domain.shift, summation_domain.basis,
                                    summation_domain.shift, claimed_sum, out.data_ptr())
        else:
            self.lib.sumcheck_g_multiplicative_dev(d_f.data_ptr(), d_h.data_ptr(), codeword_domain.dim, codeword_domain.gen, codeword_domain.shift,
                                                   summation_domain.dim, summation_domain.shift, claimed_sum, out.data_ptr())
        return out

    def lincheck(self, d_fz, d_mz, r_mz, d_p1, d_p2, n):
        out = self.empty(n)
        self.lib.lincheck_dev(d_fz.data_ptr(), [t.data_ptr() for t in d_mz], r_mz, d_p1.data_ptr(), d_p2.data_ptr(), n, out.data_ptr(),
                              prime_field=not self.field.additive)
        return out

    def ldt_combine(self, d_oracles, degrees, random_coefficients, domain):
        out = self.empty(domain.size)
        ptrs = [t.data_ptr() for t in d_oracles]
        if domain.additive:
            self.lib.ldt_combine_dev(ptrs, degrees, random_coefficients, domain.basis, domain.shift, out.data_ptr())
        else:
            self.lib.ldt_combine_multiplicative_dev(ptrs, degrees, random_coefficients, domain.dim, domain.gen, domain.shift, out.data_ptr())
        return out

    def lincomb(self, d_oracles, coefficients, n):
        out = self.empty(n)
        self.lib.lincomb_dev([t.data_ptr() for t in d_oracles], coefficients, n, out.data_ptr(), prime_field=not self.field.additive)
        return out

    # ---- vector-sized steps of the encoded prover ----
    def sub(self, d_a, d_b):
        out = self.empty(d_a.shape[0])
        if self.field.additive:
            self.lib.field_add_dev(d_a.data_ptr(), d_b.data_ptr(), out.data_ptr(), d_a.shape[0])
        else:
            self.lib.fp3_sub_dev(d_a.data_ptr(), d_b.data_ptr(), out.data_ptr(), d_a.shape[0])
        return out

    def mul(self, d_a, d_b):
        out = self.empty(d_a.shape[0])
        if self.field.additive:
            self.lib.gf192_mul_dev(d_a.data_ptr(), d_b.data_ptr(), out.data_ptr(), d_a.shape[0])
        else:
            self.lib.fp3_mul_dev(d_a.data_ptr(), d_b.data_ptr(), out.data_ptr(), d_a.shape[0])
        return out

    def inv(self, d_a):
        out = self.empty(d_a.shape[0])
        self.lib.field_inv_dev(d_a.data_ptr(), out.data_ptr(), d_a.shape[0], prime_field=not self.field.additive)
        return out

    def pow_table(self, count, base, init):
        out = self.empty(count)
        if self.field.additive:
            self.lib.pow_table_dev(out.data_ptr(), count, base, init)
        else:
            self.lib.fp3_pow_table_dev(out.data_ptr(), count, base, init)
        return out

    def spmv(self, csr, d_vec, d_out=None, scale=None, accumulate=False):
        out = self.empty(csr.rows) if d_out is None else d_out
        self.lib.spmv_dev(csr.d_row_ptr.data_ptr(), csr.d_col.data_ptr(), csr.d_coeff.data_ptr(), csr.rows, d_vec.data_ptr(), out.data_ptr(),
                          scale=scale, accumulate=accumulate, prime_field=not self.field.additive)
        return out

    # ---- vector-sized steps of the holographic prover ----
    def div(self, d_num, d_den):
        """d_num / d_den elementwise by batch inversion (d_num None: the inverses)."""
        out = self.empty(d_den.shape[0])
        self.lib.field_div_dev(d_num.data_ptr() if d_num is not None else None, d_den.data_ptr(), out.data_ptr(), d_den.shape[0],
                               prime_field=not self.field.additive)
        return out

    def domain_offsets(self, domain, point):
        """point - x over the whole domain."""
        out = self.empty(domain.size)
        if domain.additive:
            self.lib.domain_offsets_dev(domain.basis, domain.shift, point, out.data_ptr())
        else:
            self.lib.domain_offsets_multiplicative_dev(domain.dim, domain.gen, domain.shift, point, out.data_ptr())
        return out

    def domain_elements(self, domain):
        """field_subset::all_elements on the device."""
        if domain.additive:
            return self.domain_offsets(domain, self.field.zero())
        return self.pow_table(domain.size, domain.gen, domain.shift)

    def vanishing_evals(self, vanishing_domain, domain, constant):
        """constant - Z_S(x) over the whole domain (S = vanishing_domain)."""
        out = self.empty(domain.size)
        S = vanishing_domain
        if domain.additive:
            self.lib.vanishing_evals_dev(domain.basis, domain.shift, S.basis, S.shift, constant, out.data_ptr())
        else:
            self.lib.vanishing_evals_multiplicative_dev(domain.dim, domain.gen, domain.shift, S.dim, S.shift, constant, out.data_ptr())
        return out

    def lagrange_evals(self, x, S, evaldomain):
        """lagrange_polynomial(x, S, normalized = false).evaluations_over_field_subset(evaldomain) (lagrange_polynomial.tcc:66-136):
        (Z_S(x) - Z_S(y)) / (x - y) for y over evaldomain.  The reference patches the position y = x (probability |evaldomain| / |F|
        for a sampled x) with the formal derivative; here that case is refused instead of silently differing."""
        if self.field.element_in_domain(evaldomain, x):
            raise NotImplementedError("the evaluation point lies in the evaluation domain")
        numerator = self.vanishing_evals(S, evaldomain, self.field.vanishing_eval(S, x, self.lib))
        return self.div(numerator, self.domain_offsets(evaldomain, x))

    def lincomb_affine(self, d_oracles, coefficients, constant, n):
        out = self.empty(n)
        self.lib.lincomb_affine_dev([t.data_ptr() for t in d_oracles], coefficients, constant, n, out.data_ptr(), prime_field=not self.field.additive)
        return out

    def rational_combine(self, d_numerators, d_denominators, coefficients, n):
        """(combined numerator, combined denominator) of sum_i c_i N_i / D_i."""
        N, D = self.empty(n), self.empty(n)
        self.lib.rational_combine_dev([t.data_ptr() for t in d_numerators], [t.data_ptr() for t in d_denominators], coefficients, n,
                                      N.data_ptr(), D.data_ptr(), prime_field=not self.field.additive)
        return N, D

    def rational_sumcheck_constraint(self, d_p, d_N, d_D, codeword_domain, summation_domain, claimed_sum):
        out = self.empty(codeword_domain.size)
        L, K = codeword_domain, summation_domain
        if L.additive:
            if not np.array_equal(K.basis, L.basis[: K.dim]):
                raise ValueError("the summation domain must be spanned by a prefix of the codeword domain's basis")
            xinv = self.div(None, self.domain_offsets(L, self.field.zero()))
            self.lib.rational_sumcheck_constraint_dev(d_p.data_ptr(), d_N.data_ptr(), d_D.data_ptr(), xinv.data_ptr(), L.basis, L.shift, K.dim, K.shift,
                                                      claimed_sum, out.data_ptr())
        else:
            self.lib.rational_sumcheck_constraint_multiplicative_dev(d_p.data_ptr(), d_N.data_ptr(), d_D.data_ptr(), L.dim, L.gen, L.shift, K.dim, K.shift,
                                                                     claimed_sum, out.data_ptr())
        return out

    def poly_div_vanishing(self, d_poly, n_coeffs, domain, out=None):
        """polynomial_over_vanishing_polynomial(P, Z_domain).first (written to the head of `out` when given)."""
        out = self.empty(max(n_coeffs - domain.size, 0)) if out is None else out
        if n_coeffs > domain.size:
            if domain.additive:
                self.lib.poly_div_vanishing_dev(d_poly.data_ptr(), n_coeffs, domain.basis, domain.shift, out.data_ptr())
            else:
                self.lib.poly_div_vanishing_multiplicative_dev(d_poly.data_ptr(), n_coeffs, domain.dim, domain.shift, out.data_ptr())
        return out


def _p(words):
    return np.ascontiguousarray(words, dtype=np.uint64).ctypes.data_as(la._u64p)
