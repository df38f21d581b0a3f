"""FRI on device-resident codewords: the FRI-only SNARK of BASELINE config 3 on the BCS round driver (fri_snark_prover), and
the bare commit-phase loops the sharded pipelines of libiop_amd/dist.py and the stage benches reuse (fri_commit*).

The commit-phase functions are a host-side mirror of FRI_protocol::calculate_and_submit_proof (libiop/protocols/ldt/fri/fri_ldt.tcc:475-548)
together with the per-round work bcs_prover::signal_prover_round_done does for it (bcs_prover.tcc:23-60: one
Merkle tree per round over the round's oracle, leaves serialised by cosets of size 2^eta_i; bcs_common.tcc:550-614:
absorb the root, absorb the round's prover messages, squeeze the verifier challenge).  The codeword never leaves
HBM: Merkle build, fold and the final IFFT run through the C ABI's *_dev entry points on torch-owned buffers."""
import numpy as np

from . import host


class FRICommitResult:
    def __init__(self):
        self.roots = []             # one 32-byte root per round
        self.challenges = []        # x_i, (3,) uint64 each
        self.trees = []             # device node buffers (torch uint8 tensors), heap order
        self.codewords = []         # f_i device buffers (torch int64 tensors, (n_i, 3))
        self.final_polynomial = None


def fri_commit(lib, torch, d_codeword, basis, shift, localization_parameters, final_degree_bound, hashchain=None,
               keep_codewords=False, domains=None):
    """Runs the FRI reductions on `d_codeword` ((2^m, 3) int64 torch tensor on the device `lib` is bound to).
    Returns FRICommitResult.  `hashchain` defaults to a fresh Blake2bHashchain; `domains` is the chain
    host.fri_additive_domains(...) — the reference computes it once in the protocol constructor
    (FRI_protocol::compute_domains, fri_ldt.tcc:279-340), so callers that prove repeatedly pass it in."""
    hc = hashchain or host.Blake2bHashchain()
    doms = domains or host.fri_additive_domains(basis, shift, localization_parameters)
    res = FRICommitResult()
    f = d_codeword
    for i, eta in enumerate(localization_parameters):
        b_i, s_i = doms[i]
        n_i = f.shape[0]
        cs = 1 << eta
        leaves = n_i // cs
        nodes = torch.empty((2 * leaves - 1, 32), dtype=torch.uint8, device=f.device)
        # signal_prover_round_done: Merkle tree over f_i with cosets of size 2^eta_i (bcs_prover.tcc:36-46)
        lib.merkle_tree_dev([f.data_ptr()], 24, n_i, cs, nodes.data_ptr())
        root = lib.read_digest(nodes.data_ptr())        # merkle_tree::get_root, on the library's stream
        res.roots.append(root)
        res.trees.append(nodes)
        if keep_codewords:
            res.codewords.append(f)
        hc.absorb(root)             # run_hashchain_for_round: roots, then the (empty) prover messages
        hc.absorb(None)
        x_i = hc.squeeze_gf192(1)[0]
        res.challenges.append(x_i)
        nxt = torch.empty((n_i // cs, 3), dtype=torch.int64, device=f.device)
        lib.fri_fold_dev(f.data_ptr(), b_i, s_i, cs, x_i, nxt.data_ptr())     # fri_ldt.tcc:522-526
        f = nxt
    b_l, s_l = doms[len(localization_parameters)]
    coeffs = torch.empty_like(f)
    lib.additive_IFFT_dev(f.data_ptr(), b_l, s_l, coeffs.data_ptr())          # fri_ldt.tcc:538
    lib.synchronize()
    res.final_polynomial = coeffs.cpu().numpy().view(np.uint64)[:final_degree_bound].copy()   # :540 resize
    return res


class FRISnarkParameters:
    """FRI_snark_parameters / FRI_iop_protocol_parameters (libiop/snark/fri_snark.hpp, protocols/fri_iop.hpp): codeword
    domain dimension, RS_extra_dimensions, the localization array (or parameter), the interactive and query repetitions the
    harness overrides FRI's own parameterisation with (fri_iop.tcc:59-73); pow parameter = dim + 3 (fri_snark.tcc:26-28,
    common_bcs_parameters.tcc:23-25)."""

    def __init__(self, codeword_domain_dim, RS_extra_dimensions, localization_parameter=2, num_interactive_repetitions=1,
                 num_query_repetitions=10, localization_parameter_array=None):
        self.codeword_domain_dim, self.RS_extra_dimensions = codeword_domain_dim, RS_extra_dimensions
        self.localization_parameters = list(localization_parameter_array) if localization_parameter_array else \
            host.localization_parameter_to_array(localization_parameter, codeword_domain_dim, RS_extra_dimensions)
        self.num_interactive_repetitions, self.num_query_repetitions = num_interactive_repetitions, num_query_repetitions
        self.pow_bits = codeword_domain_dim + 3
        self.poly_degree_bound = 1 << (codeword_domain_dim - RS_extra_dimensions)


class FRIIopProtocol:
    """FRI_iop_protocol (libiop/protocols/fri_iop.tcc:3-101): one oracle over the unshifted default codeword domain (:13), the
    LDT instance reducer with one instance over it, FRI with the given repetitions; round-0 leaves hold cosets of 2^eta_0 (:55-57).

    Reference quirk: dummy_oracle::evaluated_contents (protocols/encoded/dummy_protocol.tcc:14-32) reserves its result and then
    loops over its still-zero size, so the virtual oracle the reference hands to the LDT reducer is EMPTY and the reference's
    FRI_snark_prover folds out-of-bounds memory (no reference test runs it).  BASELINE config 3 states the intent — the FRI prover
    on a degree-2^20 Reed-Solomon codeword — so the reducer here takes the submitted oracle itself."""

    def __init__(self, IOP, params):
        from .aurora import LDTInstanceReducer
        self.IOP, self.params = IOP, params
        domain = IOP.ops.mark_codeword_domain(IOP.field.domain(1 << params.codeword_domain_dim))
        self.codeword_domain_handle = IOP.register_domain(domain)
        self.oracle = IOP.register_oracle("dummy", self.codeword_domain_handle, params.poly_degree_bound, False)
        self.LDT = LDTInstanceReducer(IOP, self.codeword_domain_handle, 1, params.poly_degree_bound)
        IOP.set_round_parameters(domain.get_subset_of_order(1 << params.localization_parameters[0]))

    def register_interactions(self):
        self.LDT.register_interactions([self.oracle], self.params.localization_parameters, self.params.num_interactive_repetitions,
                                       self.params.num_query_repetitions)

    def register_queries(self):
        self.LDT.register_queries()

    def produce_proof(self, d_codeword):                                                   # :82-89
        self.IOP.submit_oracle(self.oracle, d_codeword)
        self.IOP.signal_prover_round_done()
        self.LDT.calculate_and_submit_proof()


def fri_snark_prover(ops, params, d_poly_coeffs=None, d_codeword=None, round_hook=None):
    """FRI_snark_prover (libiop/snark/fri_snark.tcc:43-77) on the BCS round driver: the codeword (given, or the extension of
    the given coefficients, dummy_protocol.tcc:91-107) is committed, reduced and folded on the device; returns the Transcript."""
    from .bcs import BCSProver
    IOP = BCSProver(ops, params.pow_bits)
    if round_hook is not None:
        IOP.round_hooks.append(round_hook)
    protocol = FRIIopProtocol(IOP, params)
    protocol.register_interactions()
    IOP.seal_interaction_registrations()
    protocol.register_queries()
    IOP.seal_query_registrations()
    if d_codeword is None:
        d_codeword = ops.FFT(d_poly_coeffs, d_poly_coeffs.shape[0], IOP.get_domain(protocol.codeword_domain_handle))
    protocol.produce_proof(d_codeword)
    transcript = IOP.get_transcript()
    IOP.release()
    return transcript


def squeeze_edwards_fr(hc):
    """One element of the 181-bit prime field from the hashchain: the reference's Fp extractor (blake2b.tcc:187-227: keyed
    BLAKE2b into mont_repr, bits above the modulus MSB cleared, retry with the next key until below p)."""
    import hashlib
    import libiop_amd as la
    P = la.EDWARDS_FR_MODULUS
    hc.squeeze_index += 1
    msg = hc.state + hc.squeeze_index.to_bytes(8, "little")
    key = 0
    while True:
        raw = int.from_bytes(hashlib.blake2b(msg, digest_size=24, key=key.to_bytes(8, "little")).digest(), "little")
        raw &= (1 << P.bit_length()) - 1
        if raw < P:
            break
        key += 1
    return np.array([(raw >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)], dtype=np.uint64)


def fri_commit_multiplicative(lib, torch, d_codeword, log_n, shift_int, localization_parameters, final_degree_bound, hashchain=None,
                              keep_codewords=False):
    """The same commit phase over multiplicative cosets of the 181-bit prime field (edwards_Fr): domain chain
    size >>= eta, shift <- shift^(2^eta) (fri_ldt.tcc:292-308), cosets {j + k * n / 2^eta} (subgroup.tcc:175-197),
    Merkle leaves serialised with the multiplicative position map.  `shift_int` is the canonical integer value of
    the codeword coset's shift.  Challenges: the reference's Fp extractor (blake2b.tcc:187-227: keyed BLAKE2b into
    mont_repr, bits above the modulus MSB cleared, retry with key += num_elements until < p)."""
    import hashlib
    import libiop_amd as la
    hc = hashchain or host.Blake2bHashchain()
    P = la.EDWARDS_FR_MODULUS
    res = FRICommitResult()
    f, logn, sh = d_codeword, log_n, shift_int % P
    for eta in localization_parameters:
        n_i, cs = 1 << logn, 1 << eta
        nodes = torch.empty((2 * (n_i // cs) - 1, 32), dtype=torch.uint8, device=f.device)
        lib.merkle_tree_dev([f.data_ptr()], 24, n_i, cs, nodes.data_ptr(), domain_type=la.DOMAIN_MULTIPLICATIVE)
        root = lib.read_digest(nodes.data_ptr())        # merkle_tree::get_root, on the library's stream
        res.roots.append(root)
        res.trees.append(nodes)
        if keep_codewords:
            res.codewords.append(f)
        hc.absorb(root)
        hc.absorb(None)
        x_i = squeeze_edwards_fr(hc)
        res.challenges.append(x_i)
        nxt = torch.empty((n_i // cs, 3), dtype=torch.int64, device=f.device)
        lib._check(lib.c.iopx_fri_fold_mul_fp3_dev(f.data_ptr(), logn, la._as_u64(la.edwards_subgroup_generator(logn)).ctypes.data_as(la._u64p),
                                                   la.edwards_to_montgomery([sh]).ctypes.data_as(la._u64p), cs,
                                                   x_i.ctypes.data_as(la._u64p), nxt.data_ptr()))
        f, logn = nxt, logn - eta
        sh = pow(sh, cs, P)
    coeffs = torch.empty_like(f)
    lib._check(lib.c.iopx_mul_ifft_fp3_dev(f.data_ptr(), logn, la._as_u64(la.edwards_subgroup_generator(logn)).ctypes.data_as(la._u64p),
                                           la.edwards_to_montgomery([sh]).ctypes.data_as(la._u64p), coeffs.data_ptr()))
    lib.synchronize()
    res.final_polynomial = coeffs.cpu().numpy().view(np.uint64)[:final_degree_bound].copy()
    return res
