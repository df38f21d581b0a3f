"""TEST INFRASTRUCTURE ONLY: an independent FRI verifier built on the oracle (no product code): replays the hashchain,
checks the proof of work, re-derives the query positions, authenticates every queried coset against its round's Merkle
root (validate_set_membership_proof, merkle_tree.tcc:338-483), checks each fold with the verifier-side single-coset
interpolation (additive_evaluate_next_f_i_at_coset, fri_aux.tcc:270-303) and the last one against the final polynomial
(FRI_protocol::verifier_predicate, fri_ldt.tcc:551-690)."""
import numpy as np

import oracle


def _domains(basis, shift, loc):
    """fri_ldt.tcc:310-338 through the oracle: [(basis_i, shift_i)]"""
    return [(np.asarray(basis, dtype=np.uint64), np.asarray(shift, dtype=np.uint64))] + oracle.fri_domains_additive(basis, shift, loc)


def _element(basis, shift, idx):
    """utils.tcc:8-30 ordering: shift + sum of the basis vectors selected by the bits of idx"""
    r = np.array(shift, dtype=np.uint64).copy()
    for k in range(basis.shape[0]):
        if idx >> k & 1:
            r ^= basis[k]
    return r


def verify(proof, basis, shift, loc, final_degree_bound, num_queries, pow_bitlen):
    m = basis.shape[0]
    doms = _domains(basis, shift, loc)
    hc = oracle.Hashchain()
    xs = []
    if len(proof.roots) != len(loc):
        return False, "round count"
    for root in proof.roots:
        hc.absorb(root)
        hc.absorb(bytes(32))
        xs.append(hc.squeeze(1, 3)[0])
    hc.absorb(bytes(32))
    challenge = oracle.blake2b(hc.squeeze(1, 3).tobytes(), 32)
    if not oracle.pow_verify_blake2b(challenge, proof.proof_of_work, pow_bitlen):
        return False, "proof of work"
    hc.absorb(proof.proof_of_work)
    positions = [hc.squeeze_query_positions(1, 1 << m)[0] for _ in range(num_queries)]
    if proof.final_polynomial.shape[0] > final_degree_bound:
        return False, "final polynomial degree"
    shift_bits = 0
    # value each query carries into the next round: position -> folded value
    carried = {}
    for i, eta in enumerate(loc):
        b_i, s_i = doms[i]
        n_i, cs = 1 << b_i.shape[0], 1 << eta
        prev_bits, shift_bits = shift_bits, shift_bits + eta
        leaves = sorted(set(p >> shift_bits for p in positions))
        if list(proof.leaf_positions[i]) != leaves:
            return False, "leaf positions of round %d" % i
        vals = proof.query_responses[i]
        # leaf hash = BLAKE2b of the coset's values in position order (one oracle: merkle_tree.tcc:127-134)
        leaf_hashes = np.stack([np.frombuffer(oracle.blake2b(vals[k].tobytes(), 32), dtype=np.uint8) for k in range(len(leaves))])
        try:
            ok = oracle.membership_proof_validate(proof.roots[i], n_i // cs, leaves, leaf_hashes, proof.membership_proofs[i])
        except AssertionError:
            return False, "membership proof of round %d not consumed" % i
        if not ok:
            return False, "membership proof of round %d" % i
        nxt = {}
        for k, leaf in enumerate(leaves):
            # consistency with the previous round's fold at the positions the queries came from
            for p in positions:
                if p >> shift_bits == leaf and i > 0:
                    pos_i = p >> prev_bits
                    if not np.array_equal(vals[k][pos_i - leaf * cs], carried[pos_i]):
                        return False, "fold consistency entering round %d" % i
            coset_shift = _element(b_i, s_i, leaf * cs)
            nxt[leaf] = oracle.fri_fold_at_coset(vals[k], b_i[:eta], coset_shift, xs[i])
        carried = nxt
    b_l, s_l = doms[len(loc)]
    for leaf, v in carried.items():
        point = _element(b_l, s_l, leaf)
        if not np.array_equal(oracle.poly_eval(proof.final_polynomial, point), v):
            return False, "final polynomial"
    return True, "accept"


# ---- multiplicative cosets of the 181-bit prime field -------------------------------------------------------------------
def _squeeze_fp(hc):
    """blake2b.tcc:187-227 on the oracle's own BLAKE2b: keyed hash of state || index into mont_repr, bits above the modulus MSB
    cleared, next key until the value is below p."""
    P = oracle.EDWARDS_R
    hc.idx.value += 1
    msg = bytes(hc.state) + int(hc.idx.value).to_bytes(8, "little")
    key = 0
    while True:
        raw = int.from_bytes(oracle.blake2b(msg, 24, key.to_bytes(8, "little")), "little") & ((1 << P.bit_length()) - 1)
        if raw < P:
            return np.array([(raw >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)], dtype=np.uint64)
        key += 1


def verify_multiplicative(proof, log_n, shift_int, loc, final_degree_bound, num_queries, pow_bitlen):
    P = oracle.EDWARDS_R
    hc = oracle.Hashchain()
    xs = []
    if len(proof.roots) != len(loc):
        return False, "round count"
    for root in proof.roots:
        hc.absorb(root)
        hc.absorb(bytes(32))
        xs.append(_squeeze_fp(hc))
    hc.absorb(bytes(32))
    challenge = oracle.blake2b(_squeeze_fp(hc).tobytes(), 32)
    if not oracle.pow_verify_blake2b(challenge, proof.proof_of_work, pow_bitlen):
        return False, "proof of work"
    hc.absorb(proof.proof_of_work)
    positions = [hc.squeeze_query_positions(1, 1 << log_n)[0] for _ in range(num_queries)]
    if proof.final_polynomial.shape[0] > final_degree_bound:
        return False, "final polynomial degree"
    logn, sh = log_n, shift_int % P
    carried = {}
    for i, eta in enumerate(loc):
        n_i, cs = 1 << logn, 1 << eta
        num_leaves = n_i // cs
        prev = positions
        positions = [p % num_leaves for p in prev]
        leaves = sorted(set(positions))
        if list(proof.leaf_positions[i]) != leaves:
            return False, "leaf positions of round %d" % i
        vals = proof.query_responses[i]
        leaf_hashes = np.stack([np.frombuffer(oracle.blake2b(vals[k].tobytes(), 32), dtype=np.uint8) for k in range(len(leaves))])
        try:
            ok = oracle.membership_proof_validate(proof.roots[i], num_leaves, leaves, leaf_hashes, proof.membership_proofs[i])
        except AssertionError:
            return False, "membership proof of round %d not consumed" % i
        if not ok:
            return False, "membership proof of round %d" % i
        g_i = oracle.fp_subgroup_generator(n_i)                   # generator of L^(i)'s subgroup
        g_coset = oracle.fp_subgroup_generator(cs)                # order-2^eta subgroup the cosets are shifts of
        shift_w = oracle.fp_from_ints([sh])[0]
        nxt = {}
        for k, leaf in enumerate(leaves):
            if i > 0:
                for p in prev:                                    # positions in L^(i): element p sits at slot p // num_leaves of leaf p % num_leaves
                    if p % num_leaves == leaf and not np.array_equal(vals[k][p // num_leaves], carried[p]):
                        return False, "fold consistency entering round %d" % i
            h = oracle.fp_mul(shift_w[None, :], oracle.fp_pow(g_i, leaf)[None, :])[0]
            nxt[leaf] = oracle.fp_fri_fold_at_coset(vals[k], g_coset, h, xs[i])
        carried = nxt
        logn -= eta
        sh = pow(sh, cs, P)
    g_l = oracle.fp_subgroup_generator(1 << logn)
    shift_w = oracle.fp_from_ints([sh])[0]
    for leaf, v in carried.items():
        point = oracle.fp_mul(shift_w[None, :], oracle.fp_pow(g_l, leaf)[None, :])[0]
        if not np.array_equal(oracle.fp_poly_eval(proof.final_polynomial, point), v):
            return False, "final polynomial"
    return True, "accept"
