"""Shared cases: the provers on GENERAL constraint systems — what a caller hands to iopx_aurora_instance_create — against the oracle provers fed
the same CSR triples and assignment (oracle.aurora_prove_csr / fractal_prove_csr).  tests/r1cs_general.py builds the instances: multi-term rows,
constant-column terms, repeated columns, empty rows, hot columns, arbitrary coefficients in A and B (the reference's own example has one unit
term per row, relations/examples/r1cs_examples.tcc:38-64).  Reference walks being matched: relations/r1cs.tcc:236-268 (Az, Bz, Cz),
protocols/encoded/lincheck/basic_lincheck_aux.tcc:64-88 (the transposed accumulation), protocols/encoded/lincheck/common.tcc:5-38 and
protocols/encoded/r1cs_rs_iop/fractal_indexer.tcc:47-121 (Fractal).  Used by tests/test_general_r1cs_emu.py (kernel sources compiled for the CPU)
and tests/test_gpu_general_r1cs.py (the MI355X)."""
import numpy as np

import oracle
import r1cs_general as rg
from libiop_amd import aurora, domains, fractal, r1cs

FIELDS = {"gf192": (oracle.FIELD_GF192, 0, domains.GF192), "edwards_Fr": (oracle.FIELD_EDWARDS, 1, domains.EdwardsFr)}


def python_cs(ops, inst):
    """The instance as the Python provers' device-resident R1CS."""
    mats = [r1cs.CSRMatrix(ops, rp.astype(np.int64), col.astype(np.int32), coeff, inst.num_constraints) for rp, col, coeff in inst.matrices]
    z = inst.assignment
    return r1cs.R1CS(mats[0], mats[1], mats[2], inst.num_inputs, inst.num_variables), z[:inst.num_inputs].copy(), z[inst.num_inputs:].copy()


def first_difference(mine, ref):
    return next((i for i, (a, b) in enumerate(zip(mine, ref)) if a != b), min(len(mine), len(ref)))


def native_aurora(lib, native_code, inst, monkeypatch, head_eval):
    if head_eval:
        monkeypatch.delenv("IOPX_HEAD_EVAL", raising=False)
    else:
        monkeypatch.setenv("IOPX_HEAD_EVAL", "0")
    h = lib.aurora_instance(native_code, inst.matrices, inst.num_variables, inst.num_inputs, inst.assignment)
    try:
        lib.profile_begin()
        return lib.aurora_prove(h), lib.profile_report()
    finally:
        lib.aurora_instance_free(h)


def native_fractal(lib, native_code, inst, monkeypatch, head_eval):
    if head_eval:
        monkeypatch.delenv("IOPX_HEAD_EVAL", raising=False)
    else:
        monkeypatch.setenv("IOPX_HEAD_EVAL", "0")
    h = lib.aurora_instance(native_code, inst.matrices, inst.num_variables, inst.num_inputs, inst.assignment)
    try:
        roots = lib.fractal_index(h)
        return lib.fractal_prove(h), roots
    finally:
        lib.aurora_instance_free(h)


def check_spmv(lib, torch, device, field_name, num_constraints, num_variables, seed):
    """iopx_spmv_*_dev directly: Az, Bz, Cz of a general instance against the oracle's create_Az_Bz_Cz_from_variable_assignment (r1cs.tcc:236-268),
    then the scaled / accumulating form set_challenge uses (out += scale * M v)."""
    code, _, cls = FIELDS[field_name]
    ops = domains.DeviceOps(lib, torch, device, cls())
    inst = rg.generate(field_name, num_constraints, num_variables, min(15, num_variables), seed)
    bad, az, bz, cz = oracle.r1cs_check_csr(code, inst.matrices, inst.num_variables, inst.num_inputs, inst.assignment)
    assert bad == 0
    cs, primary, auxiliary = python_cs(ops, inst)
    d_z = ops.upload(inst.F.words(inst.z))
    for M, ref in ((cs.A, az), (cs.B, bz), (cs.C, cz)):
        assert np.array_equal(ops.download(ops.spmv(M, d_z)), ref)
    scale = inst.F.rand(__import__("random").Random(seed))
    out = ops.spmv(cs.A, d_z)
    ops.spmv(cs.B, d_z, d_out=out, scale=inst.F.words([scale])[0], accumulate=True)
    expect = inst.F.words([inst.F.add(rg._dot(inst.F, inst.rows[0][i], inst.z), inst.F.mul(scale, rg._dot(inst.F, inst.rows[1][i], inst.z)))
                           for i in range(num_constraints)])
    assert np.array_equal(ops.download(out), expect)


def check_aurora(lib, torch, device, monkeypatch, field_name, num_constraints, num_variables, num_inputs, seed, python_prover=True):
    """Satisfied general instance: native prover (both schedules) and the Python prover byte-equal to the oracle prover; its verifier accepts."""
    code, native_code, cls = FIELDS[field_name]
    inst = rg.generate(field_name, num_constraints, num_variables, num_inputs, seed)
    args = (code, inst.matrices, inst.num_variables, inst.num_inputs)
    assert oracle.r1cs_check_csr(*args, inst.assignment)[0] == 0
    ref = oracle.aurora_prove_csr(*args, inst.assignment)
    assert oracle.aurora_verify_csr(*args, inst.assignment[:num_inputs], ref)
    head, prof = native_aurora(lib, native_code, inst, monkeypatch, True)
    assert head == ref, "native prover (head schedule) differs from the oracle at byte %d" % first_difference(head, ref)
    assert prof["k_count_mismatch_words"][0] == 1 and sum(v[0] for k, v in prof.items() if k.startswith("k_ldt_combine")) == 2   # no fallback taken
    whole, _ = native_aurora(lib, native_code, inst, monkeypatch, False)
    assert whole == ref, "native prover (reference schedule) differs from the oracle at byte %d" % first_difference(whole, ref)
    if python_prover:
        ops = domains.DeviceOps(lib, torch, device, cls())
        cs, primary, auxiliary = python_cs(ops, inst)
        params = aurora.AuroraParameters(ops.field, num_constraints, num_variables, num_inputs)
        mine = aurora.aurora_snark_prover(ops, cs, primary, auxiliary, params).serialize()
        assert mine == ref, "Python prover differs from the oracle at byte %d" % first_difference(mine, ref)
    return inst, ref


def check_aurora_unsatisfied(lib, torch, device, monkeypatch, field_name, num_constraints, num_variables, num_inputs, seed, kind, python_prover=True):
    """One violated constraint / one wrong primary input / one wrong auxiliary variable: the reference still emits a transcript; the ORACLE
    defines its bytes, the provers must reproduce them on either schedule, and the oracle's verifier rejects them."""
    code, native_code, cls = FIELDS[field_name]
    good = rg.generate(field_name, num_constraints, num_variables, num_inputs, seed)
    inst = rg.perturbed(good, kind, seed + 1)
    args = (code, inst.matrices, inst.num_variables, inst.num_inputs)
    violated = oracle.r1cs_check_csr(*args, inst.assignment)[0]
    assert violated >= 1 and (kind != "constraint" or violated == 1)
    ref = oracle.aurora_prove_csr(*args, inst.assignment)
    assert not oracle.aurora_verify_csr(*args, inst.assignment[:num_inputs], ref)
    head, prof = native_aurora(lib, native_code, inst, monkeypatch, True)
    assert head == ref, "native prover (head schedule, %s) differs from the oracle at byte %d" % (kind, first_difference(head, ref))
    whole, _ = native_aurora(lib, native_code, inst, monkeypatch, False)
    assert whole == ref
    if python_prover:
        ops = domains.DeviceOps(lib, torch, device, cls())
        cs, primary, auxiliary = python_cs(ops, inst)
        params = aurora.AuroraParameters(ops.field, num_constraints, num_variables, num_inputs)
        assert aurora.aurora_snark_prover(ops, cs, primary, auxiliary, params).serialize() == ref
    # the head schedule noticed on its confirmation window and fell back to the reference's schedule: head, window, then the whole domain
    assert prof["k_count_mismatch_words"][0] == 1 and sum(v[0] for k, v in prof.items() if k.startswith("k_ldt_combine")) == 3, prof
    return prof


def check_fractal(lib, torch, device, monkeypatch, field_name, num_constraints, num_inputs, seed, max_nnz=None, python_prover=True, kind=None):
    """Fractal on a general square instance with at most max_nnz (default |H|) entries per matrix: index roots and transcript of the native
    indexer / prover (both schedules) and of the Python ones equal the oracle's; kind = an unsatisfied variant (rejected by the oracle verifier)."""
    code, native_code, cls = FIELDS[field_name]
    inst = rg.generate(field_name, num_constraints, num_constraints - 1, num_inputs, seed, max_nnz=max_nnz or num_constraints)
    if kind:
        inst = rg.perturbed(inst, kind, seed + 1)
    args = (code, inst.matrices, inst.num_variables, inst.num_inputs)
    assert (oracle.r1cs_check_csr(*args, inst.assignment)[0] == 0) == (kind is None)
    ref, ref_roots = oracle.fractal_prove_csr(*args, inst.assignment)
    assert oracle.fractal_verify_csr(*args, inst.assignment[:num_inputs], ref, ref_roots) == (kind is None)
    for head_eval in (True, False):
        t, roots = native_fractal(lib, native_code, inst, monkeypatch, head_eval)
        assert roots == ref_roots, "native index roots differ (head_eval=%s)" % head_eval
        assert t == ref, "native Fractal prover (head_eval=%s, %s) differs from the oracle at byte %d" % (head_eval, kind, first_difference(t, ref))
    if python_prover:
        ops = domains.DeviceOps(lib, torch, device, cls())
        cs, primary, auxiliary = python_cs(ops, inst)
        params = fractal.FractalParameters(ops.field, cs)
        index, (roots, _) = fractal.fractal_snark_indexer(ops, cs, params)
        assert [bytes(r) for r in roots] == ref_roots
        mine = fractal.fractal_snark_prover(ops, index, cs, primary, auxiliary, params).serialize()
        assert mine == ref, "Python Fractal prover differs from the oracle at byte %d" % first_difference(mine, ref)
    return inst
