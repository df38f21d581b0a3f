"""Shared Fractal cases: the device-path indexer and prover (libiop_amd/fractal.py over the C ABI) against the oracle's independent
indexer, prover and verifier (oracle/fractal.hpp).  Used by tests/test_fractal_emu.py (kernel sources compiled for the CPU) and
the `-m gpu` tests (the real library on the MI355X)."""
import numpy as np

import oracle
from libiop_amd import domains, fractal, r1cs

FIELDS = {"gf192": (oracle.FIELD_GF192, domains.GF192), "edwards_Fr": (oracle.FIELD_EDWARDS, domains.EdwardsFr)}


def device_index_and_prove(lib, torch, device, field_name, log_n, num_inputs, seed, rs_extra=3, localization=2):
    field = FIELDS[field_name][1]()
    ops = domains.DeviceOps(lib, torch, device, field)
    n = 1 << log_n
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, seed)
    params = fractal.FractalParameters(field, cs, RS_extra_dimensions=rs_extra, FRI_localization_parameter=localization)
    prover_index, verifier_index = fractal.fractal_snark_indexer(ops, cs, params)
    transcript = fractal.fractal_snark_prover(ops, prover_index, cs, primary, auxiliary, params)
    return transcript, verifier_index, prover_index, params, ops


def check_transcript_equals_oracle(lib, torch, device, field_name, log_n, num_inputs, seed, rs_extra=3, localization=2):
    """The device index's Merkle root and the device transcript equal the oracle indexer's / prover's byte for byte, and the
    oracle's verifier accepts the transcript against the device's verifier index."""
    code = FIELDS[field_name][0]
    transcript, (roots, messages), _, params, _ = device_index_and_prove(lib, torch, device, field_name, log_n, num_inputs, seed, rs_extra, localization)
    assert messages == []
    ref, ref_roots = oracle.fractal_prove(code, log_n, num_inputs, seed, rs_extra=rs_extra, localization=localization)
    assert [bytes(r) for r in roots] == ref_roots, "index Merkle roots differ"
    mine = transcript.serialize()
    if mine != ref:
        first = next((i for i, (a, b) in enumerate(zip(mine, ref)) if a != b), min(len(mine), len(ref)))
        raise AssertionError("device transcript differs from the oracle prover's at byte %d (lengths %d / %d)" % (first, len(mine), len(ref)))
    assert oracle.fractal_verify(code, log_n, num_inputs, seed, mine, [bytes(r) for r in roots], rs_extra=rs_extra, localization=localization)
    return transcript, [bytes(r) for r in roots], params


def check_index_oracles(lib, torch, device, field_name, log_n, num_inputs, seed):
    """Each of the twelve index oracles over the codeword domain equals the oracle's matrix_indexer output."""
    code = FIELDS[field_name][0]
    field = FIELDS[field_name][1]()
    ops = domains.DeviceOps(lib, torch, device, field)
    n = 1 << log_n
    cs, _, _ = r1cs.generate_r1cs_example(ops, n, num_inputs, n - 1, seed)
    params = fractal.FractalParameters(field, cs)
    prover_index, _ = fractal.fractal_snark_indexer(ops, cs, params)
    for matrix in range(3):
        for which in range(4):
            ref = oracle.fractal_index_oracle(code, log_n, num_inputs, seed, matrix, which)
            got = ops.download(prover_index.oracles[4 * matrix + which])
            assert np.array_equal(got, ref), (matrix, which)


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

    def flip_index_response(t):
        t.query_responses[0] = t.query_responses[0].copy(); t.query_responses[0][0, 2, 0] ^= np.uint64(1)

    def flip_t_response(t):
        t.query_responses[2] = t.query_responses[2].copy(); t.query_responses[2][0, 0, 0] ^= np.uint64(1)

    def flip_claimed_value(t):
        t.prover_messages[0] = t.prover_messages[0].copy(); t.prover_messages[0][0, 0] ^= np.uint64(1)

    def flip_final(t):
        t.prover_messages[-1] = t.prover_messages[-1].copy(); t.prover_messages[-1][0, 0] ^= np.uint64(1)

    def flip_aux(t):
        t.MT_set_membership_proofs[0] = t.MT_set_membership_proofs[0].copy(); t.MT_set_membership_proofs[0][0, 0] ^= 1

    def flip_pow(t):
        p = bytearray(t.proof_of_work); p[31] ^= 0x40; t.proof_of_work = bytes(p)

    variant("round-2 root", flip_root)
    variant("index oracle answer", flip_index_response)
    variant("lincheck t answer", flip_t_response)
    variant("M(alpha, beta)", flip_claimed_value)
    variant("final polynomial", flip_final)
    variant("index authentication path", flip_aux)
    variant("proof of work", flip_pow)
    return out


# ---- the kernels behind the holographic virtual oracles, one by one, against plain field arithmetic on the host ----
def _rand(field, rng, n):
    if field.additive:
        return rng.integers(0, 1 << 63, size=(n, 3), dtype=np.uint64)
    return np.stack([field.from_int(int(rng.integers(0, 1 << 62)) ** 3 + 1) for _ in range(n)])


def check_div_kernel(lib, torch, device, field_name, n):
    """num / den by batch inversion: q * den == num, inv * den == 1, a zero denominator yields zero."""
    field = FIELDS[field_name][1]()
    ops = domains.DeviceOps(lib, torch, device, field)
    rng = np.random.default_rng(n)
    num, den = _rand(field, rng, n), _rand(field, rng, n)
    if n > 3:
        den[3] = 0
    d_num, d_den = ops.upload(num), ops.upload(den)
    q, inv = ops.div(d_num, d_den), ops.div(None, d_den)
    back, ones = ops.download(ops.mul(q, d_den)), ops.download(ops.mul(inv, d_den))
    q, inv = ops.download(q), ops.download(inv)
    expect_back, expect_ones = num.copy(), np.broadcast_to(field.one(), (n, 3)).copy()
    if n > 3:
        assert not q[3].any() and not inv[3].any()
        expect_back[3] = 0
        expect_ones[3] = 0
    assert np.array_equal(back, expect_back) and np.array_equal(ones, expect_ones)


def check_domain_kernels(lib, torch, device, field_name, log_l, log_h, samples=(0, 1, 2, 17, 255, 256)):
    """point - x, point - Z_H(x) and the unnormalised Lagrange polynomial over a shifted codeword domain, at sampled positions."""
    field = FIELDS[field_name][1]()
    ops = domains.DeviceOps(lib, torch, device, field)
    L = field.domain(1 << log_l, field.domain(1 << log_l).element_outside_of_subset())
    H = field.domain(1 << log_h)
    xs = ops.download(ops.domain_elements(L))
    point = _rand(field, np.random.default_rng(1), 1)[0]
    offs = ops.download(ops.domain_offsets(L, point))
    van = ops.download(ops.vanishing_evals(H, L, point))
    lag = ops.download(ops.lagrange_evals(point, H, L))
    z_at_point = field.vanishing_eval(H, point, lib)
    for j in [j for j in samples if j < L.size] + [L.size - 1, L.size // 2, L.size // 2 + 5]:
        assert np.array_equal(offs[j], field.sub(point, xs[j]))
        assert np.array_equal(van[j], field.sub(point, field.vanishing_eval(H, xs[j], lib)))
        expect = field.mul(field.sub(z_at_point, field.vanishing_eval(H, xs[j], lib)), field.inv(field.sub(point, xs[j]), lib))
        assert np.array_equal(lag[j], expect)
    try:
        ops.lagrange_evals(xs[5], H, L)                          # the reference patches this position; the device path refuses
    except NotImplementedError:
        return
    raise AssertionError("an evaluation point inside the evaluation domain must be refused")
