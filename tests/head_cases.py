"""Shared cases for the native provers' two schedules (head of the codeword domain / whole domain): used by tests/test_head_eval_emu.py
(kernel sources compiled for the CPU) and tests/test_gpu_head_eval.py (the real library on the MI355X)."""
import numpy as np

import aurora_cases as ac
import oracle
from libiop_amd import aurora, domains, r1cs


def _code(field_name):
    return 0 if field_name == "gf192" else 1


def prove(lib, protocol, field_name, log_n, num_inputs, seed):
    n = 1 << log_n
    inst = lib.aurora_example_instance(_code(field_name), n, num_inputs, n - 1, seed)
    try:
        roots = lib.fractal_index(inst) if protocol == "fractal" else []
        lib.profile_begin()
        t = lib.fractal_prove(inst) if protocol == "fractal" else lib.aurora_prove(inst)
        return t, roots, lib.profile_report()
    finally:
        lib.aurora_instance_free(inst)


def ldt_bytes(prof):
    return sum(v[2] for k, v in prof.items() if k.startswith("k_ldt_combine"))


def check_both_schedules(lib, monkeypatch, protocol, field_name, log_n, num_inputs):
    code = ac.FIELDS[field_name][0]
    seed = 0x2204 if protocol == "aurora" else 0x2205
    ref, ref_roots = (oracle.aurora_prove(code, log_n, num_inputs, seed), []) if protocol == "aurora" else oracle.fractal_prove(code, log_n, num_inputs, seed)
    monkeypatch.delenv("IOPX_HEAD_EVAL", raising=False)
    head, roots, head_prof = prove(lib, protocol, field_name, log_n, num_inputs, seed)
    monkeypatch.setenv("IOPX_HEAD_EVAL", "0")
    whole, roots0, whole_prof = prove(lib, protocol, field_name, log_n, num_inputs, seed)
    assert head == ref and whole == ref and roots == ref_roots and roots0 == ref_roots
    # the head schedule really ran: the LDT combination touched a fraction of the codeword domain, and the confirmation compared two windows
    assert "k_count_mismatch_words" in head_prof and "k_count_mismatch_words" not in whole_prof
    assert ldt_bytes(head_prof) * 4 <= ldt_bytes(whole_prof), (ldt_bytes(head_prof), ldt_bytes(whole_prof))


def csr(ops, M):
    return (M.row_ptr, M.col, ops.download(M.d_coeff))


def check_unsatisfied_witness(lib, torch, device, monkeypatch, field_name, log_n=8):
    """iopx_aurora_instance_create on the seeded constraint system with one auxiliary variable changed: Az * Bz != Cz, the rowcheck oracle is
    not a polynomial.  The oracle prover on the same CSR triples and assignment (oracle.aurora_prove_csr) defines the bytes; the second prover
    (libiop_amd/aurora.py: the reference's schedule) and the native prover with and without IOPX_HEAD_EVAL=0 must produce them, and the oracle
    verifier rejects them."""
    code, cls = ac.FIELDS[field_name]
    field, k, seed = cls(), 15, 0x2204
    n = 1 << log_n
    ops = domains.DeviceOps(lib, torch, device, field)
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, n, k, n - 1, seed)
    params = aurora.AuroraParameters(field, n, n - 1, k)
    mats = [csr(ops, M) for M in (cs.A, cs.B, cs.C)]
    z = np.concatenate([np.asarray(primary, dtype=np.uint64).reshape(-1, 3), np.asarray(auxiliary, dtype=np.uint64).reshape(-1, 3)])

    def native(assignment):
        inst = lib.aurora_instance(_code(field_name), mats, n - 1, k, assignment)
        try:
            lib.profile_begin()
            t = lib.aurora_prove(inst)
            return t, lib.profile_report()
        finally:
            lib.aurora_instance_free(inst)

    monkeypatch.delenv("IOPX_HEAD_EVAL", raising=False)
    good, good_prof = native(z)
    assert good == oracle.aurora_prove(code, log_n, k, seed)                    # the CSR route builds the seeded instance
    assert good_prof["k_count_mismatch_words"][0] == 1

    bad_z = z.copy()
    bad_z[k + 5] = z[k + 6]                                                     # another valid field element in an auxiliary slot
    assert not np.array_equal(bad_z, z)
    expected = oracle.aurora_prove_csr(code, mats, n - 1, k, bad_z)             # the ORACLE prover defines the bytes of the unsatisfied instance
    assert expected != good
    assert aurora.aurora_snark_prover(ops, cs, bad_z[:k], bad_z[k:], params).serialize() == expected
    bad, bad_prof = native(bad_z)
    assert bad == expected, "the head schedule changed the bytes of an unsatisfied instance's transcript"
    assert not oracle.aurora_verify(code, log_n, k, seed, bad)
    # head, confirmation window (differs), then the whole domain
    launches = sum(v[0] for kk, v in bad_prof.items() if kk.startswith("k_ldt_combine"))
    assert launches == 3 and bad_prof["k_count_mismatch_words"][0] == 1, bad_prof
    monkeypatch.setenv("IOPX_HEAD_EVAL", "0")
    assert native(bad_z)[0] == expected


def check_div_by_vanishing(lib, torch, device, m, sub_dim, seed):
    """iopx_div_by_vanishing_gf192_dev against the general route (Z_S over the domain by iopx_vanishing_evals, then the batch-inversion division),
    over a shifted standard-basis domain and over a general basis; a domain that meets S is refused."""
    import ctypes
    from helpers import rand_elems
    field = domains.GF192()
    ops = domains.DeviceOps(lib, torch, device, field)
    lib.c.iopx_div_by_vanishing_gf192_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    for kind in ("standard", "general"):
        basis = oracle.standard_basis(m, 3) if kind == "standard" else rand_elems(seed + 1, m, 3)
        shift = np.array([1 << m, 0, 0], dtype=np.uint64) if kind == "standard" else rand_elems(seed + 2, 1, 3)[0]
        sub_shift = np.zeros(3, dtype=np.uint64) if kind == "standard" else rand_elems(seed + 3, 1, 3)[0]
        D = domains.Domain(field, domains.ADDITIVE, basis=basis, shift=shift)
        S = domains.Domain(field, domains.ADDITIVE, basis=basis[:sub_dim].copy(), shift=sub_shift)
        d_in = ops.upload(rand_elems(seed, 1 << m, 3))
        expect = ops.download(ops.div(d_in, ops.vanishing_evals(S, D, np.zeros(3, dtype=np.uint64))))
        out = ops.empty(1 << m)
        b, s, ss = (np.ascontiguousarray(x, dtype=np.uint64) for x in (basis, shift, sub_shift))
        lib._check(lib.c.iopx_div_by_vanishing_gf192_dev(d_in.data_ptr(), b.ctypes.data, m, s.ctypes.data, sub_dim, ss.ctypes.data, out.data_ptr()))
        assert np.array_equal(ops.download(out), expect), (kind, m, sub_dim)
    basis, zero = oracle.standard_basis(m, 3), np.zeros(3, dtype=np.uint64)
    try:
        lib._check(lib.c.iopx_div_by_vanishing_gf192_dev(d_in.data_ptr(), basis.ctypes.data, m, zero.ctypes.data, sub_dim, zero.ctypes.data, out.data_ptr()))
    except ValueError:
        return
    raise AssertionError("a domain that contains S was accepted")


def check_reextend2(lib, torch, device, m, d, batch_a, batch_b, seed, general=False):
    """iopx_add_reextend2_gf192_batch_dev against the oracle's IFFT over each group's own coset followed by the FFT over the codeword domain, on a
    coset range, for the standard basis (the prover's) and a random one."""
    from helpers import rand_elems
    field = domains.GF192()
    ops = domains.DeviceOps(lib, torch, device, field)
    basis = rand_elems(seed + 1, m, 3) if general else oracle.standard_basis(m, 3)
    shift = rand_elems(seed + 2, 1, 3)[0] if general else np.array([1 << m, 0, 0], dtype=np.uint64)
    shift_a, shift_b = np.zeros(3, dtype=np.uint64), shift.copy()                  # H itself and the first coset of its span inside L
    nd, cosets = 1 << d, 1 << (m - d)
    ev_a, ev_b = rand_elems(seed + 3, batch_a * nd, 3), rand_elems(seed + 4, batch_b * nd, 3)
    first, count = (1, cosets - 1) if cosets > 2 else (0, cosets)
    d_a, d_b = ops.upload(ev_a), ops.upload(ev_b)
    outs = [ops.empty(count * nd) for _ in range(batch_a + batch_b)]
    lib.additive_reextend2_batch_dev(d_a.data_ptr(), batch_a, shift_a, d_b.data_ptr(), batch_b, shift_b, basis, d, shift, first, count, [o.data_ptr() for o in outs])
    for k in range(batch_a + batch_b):
        ev, sh = (ev_a[k * nd:(k + 1) * nd], shift_a) if k < batch_a else (ev_b[(k - batch_a) * nd:(k - batch_a + 1) * nd], shift_b)
        coeffs = oracle.additive_ifft(ev, basis[:d], sh)
        full = oracle.additive_fft(coeffs, basis, shift)
        assert np.array_equal(ops.download(outs[k]), full[first * nd:(first + count) * nd]), (m, d, k, general)


def check_query_phase_behind_the_grind(lib, monkeypatch, protocol, field_name, log_n, num_inputs):
    code = ac.FIELDS[field_name][0]
    seed = 0x2204 if protocol == "aurora" else 0x2205
    ref = oracle.aurora_prove(code, log_n, num_inputs, seed) if protocol == "aurora" else oracle.fractal_prove(code, log_n, num_inputs, seed)[0]
    monkeypatch.setenv("IOPX_POW_BEHIND_LOG2", "0")
    behind = prove(lib, protocol, field_name, log_n, num_inputs, seed)[0]
    monkeypatch.setenv("IOPX_POW_BEHIND_LOG2", "40")
    after = prove(lib, protocol, field_name, log_n, num_inputs, seed)[0]
    assert behind == ref and after == ref
