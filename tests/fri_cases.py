"""FRI-only SNARK cases (BASELINE config 3's shape) shared by the CPU-emulation and GPU suites: the device-path prover
(libiop_amd/fri.py on the BCS round driver) against the oracle's independent prover and verifier (oracle/aurora.hpp
FRI_snark_*): byte-equal transcripts, acceptance, rejection of tampered transcripts and of a codeword far from low degree."""
import copy

import numpy as np

import oracle
from libiop_amd import domains, fri, r1cs

FIELDS = {"gf192": (oracle.FIELD_GF192, domains.GF192), "edwards_Fr": (oracle.FIELD_EDWARDS, domains.EdwardsFr)}


def prove_and_verify(lib, torch, device, field_name, dim, rs_extra, loc_param, interactions, queries, seed, tamper=True):
    code, cls = FIELDS[field_name]
    ops = domains.DeviceOps(lib, torch, device, cls())
    params = fri.FRISnarkParameters(dim, rs_extra, loc_param, interactions, queries)
    coeffs = r1cs.seeded_elements(ops.field, seed, params.poly_degree_bound)
    transcript = fri.fri_snark_prover(ops, params, d_poly_coeffs=ops.upload(coeffs))
    mine = transcript.serialize()
    ref = oracle.fri_snark_prove(code, dim, rs_extra, loc_param, interactions, queries, seed)
    if mine != ref:
        first = next((i for i, (a, b) in enumerate(zip(mine, ref)) if a != b), min(len(mine), len(ref)))
        raise AssertionError("device transcript differs from the oracle prover's at byte %d (lengths %d / %d)" % (first, len(mine), len(ref)))
    args = (code, dim, rs_extra, loc_param, interactions, queries)
    assert oracle.fri_snark_verify(*args, mine)
    if tamper:
        def variant(edit):
            t = copy.deepcopy(transcript)
            edit(t)
            return t.serialize()

        def flip_root(t):
            r = bytearray(t.MT_roots[-1]); r[0] ^= 1; t.MT_roots[-1] = bytes(r)

        def flip_answer(t):
            t.query_responses[1] = t.query_responses[1].copy(); t.query_responses[1][0, 0, 1] ^= np.uint64(1)

        def flip_final(t):
            t.prover_messages[0] = t.prover_messages[0].copy(); t.prover_messages[0][0, 0] ^= np.uint64(1)

        def flip_pow(t):
            p = bytearray(t.proof_of_work); p[31] ^= 0x80; t.proof_of_work = bytes(p)

        def flip_path(t):
            t.MT_set_membership_proofs[0] = t.MT_set_membership_proofs[0].copy(); t.MT_set_membership_proofs[0][-1, 3] ^= 1

        for label, edit in (("root", flip_root), ("answer", flip_answer), ("final polynomial", flip_final), ("proof of work", flip_pow), ("path", flip_path)):
            assert not oracle.fri_snark_verify(*args, variant(edit)), label
        # a codeword far from every polynomial of the tested degree must be rejected
        domain = ops.field.domain(1 << dim)
        far = ops.FFT(ops.upload(r1cs.seeded_elements(ops.field, seed + 1, min(4 * params.poly_degree_bound, 1 << dim))),
                      min(4 * params.poly_degree_bound, 1 << dim), domain)
        bad = fri.fri_snark_prover(ops, params, d_codeword=far).serialize()
        assert not oracle.fri_snark_verify(*args, bad), "degree 4d codeword accepted"
    return True


def native_prove_equals_oracle(lib, torch, device, field_name, dim, rs_extra, loc_param, interactions, queries, seed, comm=None):
    """FRI_snark_prover through the C ABI (libiop_amd/cpp/fri.hpp inside the library) against the oracle prover, byte for byte."""
    code, cls = FIELDS[field_name]
    ops = domains.DeviceOps(lib, torch, device, cls())
    params = fri.FRISnarkParameters(dim, rs_extra, loc_param, interactions, queries)
    coeffs = ops.upload(r1cs.seeded_elements(ops.field, seed, params.poly_degree_bound))
    mine = lib.fri_snark_prove(0 if field_name == "gf192" else 1, coeffs.data_ptr(), params.poly_degree_bound, dim, rs_extra, loc_param, interactions, queries, comm=comm)
    ref = oracle.fri_snark_prove(code, dim, rs_extra, loc_param, interactions, queries, seed)
    assert mine == ref, "native FRI SNARK transcript differs from the oracle prover's"
    assert oracle.fri_snark_verify(code, dim, rs_extra, loc_param, interactions, queries, mine)
    return True
