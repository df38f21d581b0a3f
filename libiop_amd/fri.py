"""FRI prover commit phase on device-resident codewords (additive domains, GF(2^192)).

Host-side mirror of FRI_protocol::calculate_and_submit_proof (libiop/protocols/ldt/fri/fri_ldt.tcc:475-548)
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


class FRIProof:
    """What the BCS transformation of the FRI protocol sends (bcs_common.hpp:36-106), for one oracle / one interaction:
    per round the Merkle root, the queried leaf positions, the codeword values of the queried cosets and a pruned
    set-membership proof; the final polynomial; the proof-of-work answer."""

    def __init__(self):
        self.roots = []
        self.final_polynomial = None
        self.proof_of_work = None
        self.leaf_positions = []        # per round: sorted unique leaf (coset) indices
        self.query_responses = []       # per round: (len(leaf_positions), coset_size, 3) uint64
        self.membership_proofs = []     # per round: (count, 32) uint8 auxiliary hashes


def fri_query_positions(hashchain, num_queries, domain_size):
    """One squeezed position per query repetition in the codeword domain (bcs_common.tcc:536-548)."""
    return [hashchain.squeeze_query_positions(1, domain_size)[0] for _ in range(num_queries)]


def fri_prove(lib, torch, d_codeword, basis, shift, localization_parameters, final_degree_bound, num_queries, pow_bitlen,
              domains=None):
    """Non-interactive FRI prover for one device-resident codeword over an affine subspace of GF(2^192): commit phase
    (fri_commit), proof of work on the hashchain's root-type squeeze (bcs_prover.tcc:52-59), query positions from the
    hashchain, then transcript extraction straight from the device-resident trees and codewords (bcs_prover.tcc:136-233).
    Round i's leaves are the cosets of size 2^eta_i of L^(i); a query at position s of L^(0) touches leaf s >> (eta_0 + .. +
    eta_i) of round i (fri_aux.tcc:355-387 for affine subspaces)."""
    hc = host.Blake2bHashchain()
    com = fri_commit(lib, torch, d_codeword, basis, shift, localization_parameters, final_degree_bound, hashchain=hc,
                     keep_codewords=True, domains=domains)
    proof = FRIProof()
    proof.roots = com.roots
    proof.final_polynomial = com.final_polynomial
    hc.absorb(None)                                       # the last round's prover message: the final polynomial (F8: state only)
    challenge = hc.squeeze_root_type()
    proof.proof_of_work = lib.solve_pow(challenge, pow_bitlen)
    hc.absorb(proof.proof_of_work)
    n0 = d_codeword.shape[0]
    positions = fri_query_positions(hc, num_queries, n0)
    shift_bits = 0
    for i, eta in enumerate(localization_parameters):
        shift_bits += eta
        f_i, nodes = com.codewords[i], com.trees[i]
        cs = 1 << eta
        leaves = sorted(set(p >> shift_bits for p in positions))
        elems = [leaf * cs + k for leaf in leaves for k in range(cs)]
        vals = lib.query_responses_dev([f_i.data_ptr()], 24, f_i.shape[0], elems)
        proof.leaf_positions.append(leaves)
        proof.query_responses.append(vals.reshape(len(leaves), cs, 3))
        proof.membership_proofs.append(lib.get_set_membership_proof_dev(nodes.data_ptr(), f_i.shape[0] // cs, leaves))
    return proof


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


def fri_prove_multiplicative(lib, torch, d_codeword, log_n, shift_int, localization_parameters, final_degree_bound, num_queries, pow_bitlen):
    """fri_prove over multiplicative cosets of the 181-bit prime field.  Round i's leaf j is the coset {j + k * n_i / 2^eta_i} of
    L^(i) (subgroup.tcc:175-197); a query at position p of L^(i) lands in leaf p mod (n_i / 2^eta_i), which is also its position
    in L^(i+1) (fri_aux.tcc:355-387 for multiplicative cosets)."""
    hc = host.Blake2bHashchain()
    com = fri_commit_multiplicative(lib, torch, d_codeword, log_n, shift_int, localization_parameters, final_degree_bound, hashchain=hc,
                                    keep_codewords=True)
    proof = FRIProof()
    proof.roots = com.roots
    proof.final_polynomial = com.final_polynomial
    hc.absorb(None)
    import hashlib
    challenge = hashlib.blake2b(squeeze_edwards_fr(hc).tobytes(), digest_size=32).digest()     # squeeze_root_type (blake2b.tcc:105-110)
    proof.proof_of_work = lib.solve_pow(challenge, pow_bitlen)
    hc.absorb(proof.proof_of_work)
    positions = fri_query_positions(hc, num_queries, 1 << log_n)
    for i, eta in enumerate(localization_parameters):
        f_i, nodes = com.codewords[i], com.trees[i]
        cs = 1 << eta
        num_leaves = f_i.shape[0] // cs
        positions = [p % num_leaves for p in positions]
        leaves = sorted(set(positions))
        vals = lib.query_responses_dev([f_i.data_ptr()], 24, f_i.shape[0], [leaf + k * num_leaves for leaf in leaves for k in range(cs)])
        proof.leaf_positions.append(leaves)
        proof.query_responses.append(vals.reshape(len(leaves), cs, 3))
        proof.membership_proofs.append(lib.get_set_membership_proof_dev(nodes.data_ptr(), num_leaves, leaves))
    return proof
