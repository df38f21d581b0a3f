"""Shared Aurora prover cases: the device-path prover (libiop_amd/aurora.py over the C ABI) against the oracle's independent
prover and verifier (oracle/aurora.hpp).  Used by tests/test_aurora_emu.py (kernel sources compiled for the CPU) and the
`-m gpu` tests (the real library on the MI355X)."""
import numpy as np

import oracle
from libiop_amd import aurora, domains, r1cs

FIELDS = {"gf192": (oracle.FIELD_GF192, domains.GF192), "edwards_Fr": (oracle.FIELD_EDWARDS, domains.EdwardsFr)}


def device_prove(lib, torch, device, field_name, log_n, num_inputs, seed, rs_extra=5, localization=2):
    field = FIELDS[field_name][1]()
    ops = domains.DeviceOps(lib, torch, device, field)
    n = 1 << log_n
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, seed)
    params = aurora.AuroraParameters(field, n, n - 1, num_inputs, RS_extra_dimensions=rs_extra, FRI_localization_parameter=localization)
    transcript = aurora.aurora_snark_prover(ops, cs, primary, auxiliary, params)
    return transcript, params, (cs, primary, auxiliary)


def check_transcript_equals_oracle(lib, torch, device, field_name, log_n, num_inputs, seed, rs_extra=5, localization=2):
    """The device transcript — roots, final polynomial, query positions and answers, authentication paths, proof of work —
    equals the oracle prover's byte for byte, and the oracle's verifier accepts it."""
    code = FIELDS[field_name][0]
    transcript, params, _ = device_prove(lib, torch, device, field_name, log_n, num_inputs, seed, rs_extra, localization)
    mine = transcript.serialize()
    ref = oracle.aurora_prove(code, log_n, num_inputs, seed, rs_extra=rs_extra, localization=localization)
    if mine != ref:
        first = next((i for i, (a, b) in enumerate(zip(mine, ref)) if a != b), min(len(mine), len(ref)))
        raise AssertionError("device transcript differs from the oracle prover's at byte %d (lengths %d / %d)" % (first, len(mine), len(ref)))
    assert oracle.aurora_verify(code, log_n, num_inputs, seed, mine, rs_extra=rs_extra, localization=localization)
    return transcript, params


def tamper_cases(transcript):
    """(label, bytes) of transcripts with one component changed; every one must be rejected."""
    import copy
    out = []

    def variant(label, edit):
        t = copy.deepcopy(transcript)
        edit(t)
        out.append((label, t.serialize()))

    def flip_root(t):
        r = bytearray(t.MT_roots[1]); r[5] ^= 1; t.MT_roots[1] = bytes(r)

    def flip_response(t):
        t.query_responses[0] = t.query_responses[0].copy(); t.query_responses[0][0, 1, 0] ^= np.uint64(1)

    def flip_h_response(t):
        t.query_responses[1] = t.query_responses[1].copy(); t.query_responses[1][0, 0, 0] ^= np.uint64(1)

    def flip_final(t):
        t.prover_messages[-1] = t.prover_messages[-1].copy(); t.prover_messages[-1][0, 0] ^= np.uint64(1)

    def flip_aux(t):
        t.MT_set_membership_proofs[2] = t.MT_set_membership_proofs[2].copy(); t.MT_set_membership_proofs[2][0, 0] ^= 1

    def flip_pow(t):
        p = bytearray(t.proof_of_work); p[31] ^= 0x40; t.proof_of_work = bytes(p)

    def shift_position(t):
        t.query_positions[0] = list(t.query_positions[0]); t.query_positions[0][0] ^= 2

    variant("round-1 root", flip_root)
    variant("witness oracle answer", flip_response)
    variant("sumcheck h answer", flip_h_response)
    variant("final polynomial", flip_final)
    variant("authentication path", flip_aux)
    variant("proof of work", flip_pow)
    variant("query position", shift_position)
    return out
