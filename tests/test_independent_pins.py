"""Checks that do NOT go through oracle/: pure-Python restatements (hashlib, integers) written from the reference text, applied to
the product's own outputs.  ADVICE r2 (low): every other end-to-end test compares the device prover with the in-repo oracle, which
has the same authors; these pin the pieces a shared misreading could hide behind, and hand-derive the parameter values the
reference's formulas give for the instrumented configurations."""
import hashlib

import numpy as np
import pytest

from emu_lib import emu
from helpers import rand_elems


# ---- BLAKE2b of the native prover's hashchain (libiop_amd/csrc/prover_support.hip) against hashlib (RFC 7693) ----
@pytest.mark.parametrize("digest_size", [8, 24, 32, 64])
def test_library_host_blake2b_equals_hashlib(digest_size):
    lib = emu()
    rng = np.random.default_rng(digest_size)
    for length in [0, 1, 24, 32, 40, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 1000]:
        msg = bytes(rng.integers(0, 256, size=length, dtype=np.uint8))
        for key in (b"", bytes(range(8)), bytes(rng.integers(0, 256, size=64, dtype=np.uint8))):
            want = hashlib.blake2b(msg, digest_size=digest_size, key=key).digest()
            assert lib.blake2b_host(msg, digest_size, key) == want, (length, len(key))


def test_product_library_host_blake2b_equals_hashlib():
    """The same function in the gfx950 build (host code: runs without a GPU)."""
    import libiop_amd
    lib = libiop_amd.Library()
    for msg, key in ((b"", b""), (b"abc", b""), (b" " * 32 + (7).to_bytes(8, "little"), (3).to_bytes(8, "little")), (bytes(range(200)), bytes(range(17)))):
        assert lib.blake2b_host(msg, 24, key) == hashlib.blake2b(msg, digest_size=24, key=key).digest()


# ---- hashchain semantics (bcs/hashing/blake2b.tcc:10-110) restated with hashlib, against the NATIVE prover's challenges ----
def _reference_hashchain_first_challenges(num_rounds_absorbed, sizes):
    """state_0 = 32 x 0x20 (:17); absorb: state <- BLAKE2b-256(first 32 bytes of state || input) = BLAKE2b-256(state) (:56-60);
    squeeze #q of n elements: element i = BLAKE2b(state || q_le64, key = i_le64, 24 bytes) (:76-86, :162-185, :231-257)."""
    state = b" " * 32
    for _ in range(num_rounds_absorbed):
        state = hashlib.blake2b(state, digest_size=32).digest()
    q, out = 0, []
    for n in sizes:
        q += 1
        out.append([hashlib.blake2b(state + q.to_bytes(8, "little"), digest_size=24, key=i.to_bytes(8, "little")).digest() for i in range(n)])
    return out


def test_round_challenges_are_the_restated_hashchain_output():
    """A round with one tree and no prover message absorbs twice (root, then the empty message list); the verifier messages that follow
    are squeezes 1, 2, ...: the restated bytes are compared with what libiop_amd/host.py (the Python provers' hashchain) and the
    oracle's hashchain produce — whatever was absorbed (the reference's absorb ignores its input, blake2b.tcc:56-60)."""
    import oracle
    from libiop_amd import host
    want = _reference_hashchain_first_challenges(2, [2, 1])
    hc = host.Blake2bHashchain()
    hc.absorb(b"root"); hc.absorb(None)
    got = [hc.squeeze_gf192(2), hc.squeeze_gf192(1)]
    for w, g in zip(want, got):
        assert [bytes(np.asarray(e, dtype=np.uint64).tobytes()) for e in g] == w
    oh = oracle.Hashchain()
    oh.absorb(b"another root entirely, 32 bytes.." + b"x"); oh.absorb(b"\0" * 32)
    assert [bytes(np.asarray(e, dtype=np.uint64).tobytes()) for e in oh.squeeze(2, 3)] == want[0]


# ---- Merkle set-membership proofs (bcs/merkle_tree.tcc:92-151, 242-336, 338-420) validated by a pure-Python verifier ----
def _validate(root, num_leaves, positions, leaf_digests, aux):
    """merkle_tree::validate_set_membership_proof restated: rebuild the queried nodes level by level, taking a sibling from `aux`
    exactly when it is not itself in the set (left node first), and compare the root."""
    nodes = {num_leaves - 1 + p: d for p, d in zip(positions, leaf_digests)}
    aux = list(aux)
    level = sorted(nodes)
    while level != [0]:
        nxt, i = [], 0
        while i < len(level):
            pos = level[i]
            if pos % 2 == 0:                                  # right child: its left sibling comes from the proof
                left, right = aux.pop(0), nodes[pos]
                i += 1
            elif i + 1 < len(level) and level[i + 1] == pos + 1:
                left, right = nodes[pos], nodes[pos + 1]
                i += 2
            else:
                left, right = nodes[pos], aux.pop(0)
                i += 1
            parent = (pos - 1) // 2
            nodes[parent] = hashlib.blake2b(left + right, digest_size=32).digest()
            nxt.append(parent)
        level = nxt
    return not aux and nodes[0] == root


def test_every_subset_of_an_eight_leaf_tree_validates_independently():
    """tests/bcs/test_merkle_tree.cpp:117-167 (run_multi_test): tree.construct({vec1, vec2}) over 8 positions, every subset of the
    leaves; the tree and the proofs come from the product kernels (CPU build), the leaf digests and the validation from hashlib."""
    check_every_subset_validates(emu())


def check_every_subset_validates(lib):
    """(also run on the HIP library: tests/test_gpu_parity.py::test_membership_proofs_validate_with_hashlib_only)"""
    L = 8
    vec1, vec2 = rand_elems(11, L, 3), rand_elems(12, L, 3)
    nodes = lib.merkle_tree([vec1, vec2], 1)
    leaf = [hashlib.blake2b(vec1[i].tobytes() + vec2[i].tobytes(), digest_size=32).digest() for i in range(L)]
    assert [bytes(nodes[L - 1 + i]) for i in range(L)] == leaf
    root = bytes(nodes[0])
    d = lib.malloc(nodes.nbytes)
    lib.h2d(d, nodes)
    try:
        for subset in range(1, 1 << L):
            positions = [k for k in range(L) if subset >> k & 1]
            aux = [bytes(a) for a in lib.get_set_membership_proof_dev(d, L, positions)]
            assert _validate(root, L, positions, [leaf[p] for p in positions], aux), subset
            if aux:
                bad = [bytes([aux[0][0] ^ 1]) + aux[0][1:]] + aux[1:]
                assert not _validate(root, L, positions, [leaf[p] for p in positions], bad)
    finally:
        lib.free(d)


# ---- parameters of the instrumented configurations, derived by hand from the reference's formulas ----
def test_aurora_2p20_parameters_follow_the_reference_formulas():
    """profiling/instrument_aurora_snark.cpp:95-122 over gf192, n = 2^20, non-zk: codeword dimension 20 + 5 = 25
    (aurora_iop.tcc:35-43); pow parameter 20 + 3 (common_bcs_parameters.tcc:23-25) so query soundness 128 + 1 - 23 = 106 bits;
    localization array [1] + [2] * ((25 - 5 - 1) // 2) = [1, 2 x 9] (fri_ldt.tcc:132-146); tested degree 2^20, constraint degree
    2^21 - 1 (r1cs_rs_iop.tcc:56-63); proximity min(2^25 - 2^21 + 1, 2^25 - 2^20) - 1 = 2^25 - 2^21 (ldt_reducer.tcc:34-42), i.e.
    1 - delta = 1/16 exactly, 4 bits per query: ceil(106 / 4) = 27 queries (fri_ldt.tcc:83-106); 131 interactive bits against a
    192-bit field: one repetition of everything."""
    from libiop_amd import aurora, domains
    p = aurora.AuroraParameters(domains.GF192(), 1 << 20, (1 << 20) - 1, 15)
    assert p.codeword_domain_dim == 25 and p.pow_bits == 23 and p.query_soundness_error_bits == 106
    assert p.localization_parameters == [1] + [2] * 9
    assert p.absolute_proximity_parameter == (1 << 25) - (1 << 21)
    assert p.fri_query_repetitions == 27
    assert (p.multi_lincheck_repetitions, p.num_output_LDT_instances, p.fri_interactive_repetitions) == (1, 1, 1)


def test_fractal_2p20_parameters_follow_the_reference_formulas():
    """profiling/instrument_fractal_snark.cpp:93-120 over the 181-bit field, n = 2^20, one non-zero per row: index domain 2^20,
    codeword dimension log2(4 * 2^20) + 3 = 25 (fractal_hiop.tcc:28-44), array [1, 2 x 10]; tested degree 3 * 2^20 rounded up to a
    multiple of 2^21 = 2^22 (fri_ldt.tcc:148-163), constraint degree 4 * 2^20; proximity 2^25 - 2^22 - 1, so
    1 - delta = (2^22 + 1) / 2^25, 2.99999966 bits per query: ceil(106 / 2.99999966) = 36 queries."""
    import types
    from libiop_amd import domains, fractal
    n = 1 << 20
    M = types.SimpleNamespace(row_ptr=np.array([0, n]), rows=n)
    cs = types.SimpleNamespace(A=M, B=M, C=M, num_inputs=0, num_variables=n - 1, num_constraints=lambda: n)
    p = fractal.FractalParameters(domains.EdwardsFr(), cs)
    assert (p.index_domain_dim, p.codeword_domain_dim) == (20, 25) and p.localization_parameters == [1] + [2] * 10
    assert p.max_LDT_tested_degree_bound == 1 << 22 and p.max_constraint_degree_bound == 1 << 22
    assert p.absolute_proximity_parameter == (1 << 25) - (1 << 22) - 1
    assert p.fri_query_repetitions == 36
