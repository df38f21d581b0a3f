"""Fractal indexer and prover on the device path, kernel sources compiled for the CPU (tests/emu): index oracles and index Merkle
root against the oracle's indexer, transcript byte-equality with the oracle's independent prover, acceptance by the oracle's
verifier, rejection of tampered transcripts / a wrong index / a wrong statement, parameter derivation, the new kernels one by one.
The same cases run on the MI355X in tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import emu_lib
import fractal_cases as fc
import oracle
from libiop_amd import domains, fractal, r1cs

CPU = torch.device("cpu")


@pytest.mark.parametrize("field_name,log_n,num_inputs", [("gf192", 5, 3), ("edwards_Fr", 5, 0), ("edwards_Fr", 6, 3)])
def test_index_oracles_match_oracle_indexer(field_name, log_n, num_inputs):
    fc.check_index_oracles(emu_lib.emu(), torch, CPU, field_name, log_n, num_inputs, 0x2205)


@pytest.mark.parametrize("field_name,log_n,num_inputs", [
    ("gf192", 5, 3), ("gf192", 6, 15), ("gf192", 7, 1),
    ("edwards_Fr", 5, 0), ("edwards_Fr", 6, 0), ("edwards_Fr", 7, 15), ("edwards_Fr", 8, 0), ("edwards_Fr", 9, 3),
])
def test_device_transcript_equals_oracle_prover(field_name, log_n, num_inputs):
    fc.check_transcript_equals_oracle(emu_lib.emu(), torch, CPU, field_name, log_n, num_inputs, 0x2205)


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_other_rates_and_localizations(field_name):
    # (num_inputs = 1 over multiplicative domains is reference quirk F15, see test_reference_quirk_single_input)
    fc.check_transcript_equals_oracle(emu_lib.emu(), torch, CPU, field_name, 6, 3, 7, rs_extra=2, localization=3)
    fc.check_transcript_equals_oracle(emu_lib.emu(), torch, CPU, field_name, 6, 7, 8, rs_extra=4, localization=1)


def test_reference_quirk_single_input():
    """F15: with one primary input over multiplicative domains the reference's indexer (libff::log2(num_inputs) = 0) and its
    lincheck (log2(num_inputs + 1) = 1) disagree about the column reindexing, and its own proof is rejected.  The device prover
    follows the reference: same bytes, same rejection."""
    lib = emu_lib.emu()
    transcript, (roots, _), _, _, _ = fc.device_index_and_prove(lib, torch, CPU, "edwards_Fr", 6, 1, 8)
    ref, ref_roots = oracle.fractal_prove(oracle.FIELD_EDWARDS, 6, 1, 8)
    assert [bytes(r) for r in roots] == ref_roots and transcript.serialize() == ref
    assert not oracle.fractal_verify(oracle.FIELD_EDWARDS, 6, 1, 8, ref, ref_roots)
    fc.check_transcript_equals_oracle(lib, torch, CPU, "gf192", 6, 1, 8)          # subspaces: reindexing is the identity


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_oracle_verifier_rejects_tampering(field_name):
    code = fc.FIELDS[field_name][0]
    transcript, roots, _ = fc.check_transcript_equals_oracle(emu_lib.emu(), torch, CPU, field_name, 6, 3, 11)
    for label, data in fc.tamper_cases(transcript):
        assert not oracle.fractal_verify(code, 6, 3, 11, data, roots), label
    good = transcript.serialize()
    assert not oracle.fractal_verify(code, 6, 3, 11, good[:-1], roots), "truncated"
    bad_root = bytearray(roots[0]); bad_root[0] ^= 1
    assert not oracle.fractal_verify(code, 6, 3, 11, good, [bytes(bad_root)]), "another index"
    assert not oracle.fractal_verify(code, 6, 3, 11, good, []), "no index"
    z, _, _ = oracle.r1cs_example(code, 6, 3, 11)
    wrong = z[:3].copy()
    wrong[1] = z[2]
    assert not oracle.fractal_verify(code, 6, 3, 11, good, roots, primary_override=wrong), "wrong primary input"
    assert oracle.fractal_verify(code, 6, 3, 11, good, roots, primary_override=z[:3].copy())


def test_one_index_serves_several_proofs():
    """The prover index is not consumed: two proofs of the same instance from one index are identical (deterministic prover)."""
    field = domains.EdwardsFr()
    ops = domains.DeviceOps(emu_lib.emu(), torch, CPU, field)
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, 32, 0, 31, 5)
    params = fractal.FractalParameters(field, cs)
    index, (roots, _) = fractal.fractal_snark_indexer(ops, cs, params)
    a = fractal.fractal_snark_prover(ops, index, cs, primary, auxiliary, params).serialize()
    b = fractal.fractal_snark_prover(ops, index, cs, primary, auxiliary, params).serialize()
    assert a == b
    assert oracle.fractal_verify(oracle.FIELD_EDWARDS, 5, 0, 5, a, [bytes(r) for r in roots])


@pytest.mark.parametrize("field_name,log_n,num_inputs", [("gf192", 8, 15), ("gf192", 20, 15), ("edwards_Fr", 12, 0), ("edwards_Fr", 20, 0)])
def test_parameters_match_oracle(field_name, log_n, num_inputs):
    code, cls = fc.FIELDS[field_name]

    class Shape:                                    # the parameters read sizes and non-zero counts only
        def __init__(self, n, k):
            self.num_inputs, self.num_variables = k, n - 1
            self.A = self.B = self.C = type("M", (), {"row_ptr": np.arange(n + 1)})()
            self._n = n

        def num_constraints(self):
            return self._n

    p = fractal.FractalParameters(cls(), Shape(1 << log_n, num_inputs))
    ref = oracle.fractal_params(code, log_n, num_inputs)
    for name in ("codeword_domain_dim", "pow_bits", "query_soundness_error_bits", "max_LDT_tested_degree_bound", "max_constraint_degree_bound",
                 "absolute_proximity_parameter", "holographic_lincheck_repetitions", "num_output_LDT_instances", "fri_interactive_repetitions",
                 "fri_query_repetitions", "index_domain_dim", "matrix_domain_dim", "localization_parameters"):
        assert getattr(p, name) == ref[name], name
    if log_n == 20:                                 # cfg5's shape
        assert p.codeword_domain_dim == 25 and p.localization_parameters == [1] + [2] * 10 and p.pow_bits == 23 and p.index_domain_dim == 20


def test_argument_checks():
    field = domains.EdwardsFr()
    ops = domains.DeviceOps(emu_lib.emu(), torch, CPU, field)
    cs, _, _ = r1cs.generate_r1cs_example(ops, 32, 3, 30, 5)          # not square
    with pytest.raises(ValueError):
        fractal.FractalParameters(field, cs)


# ---- the kernels behind the holographic virtual oracles, one by one, against plain field arithmetic on the host ----
@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
@pytest.mark.parametrize("n", [1, 7, 300, 2500, 9000])
def test_div_kernel(field_name, n):
    fc.check_div_kernel(emu_lib.emu(), torch, CPU, field_name, n)


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_domain_kernels(field_name):
    fc.check_domain_kernels(emu_lib.emu(), torch, CPU, field_name, 9, 4)


def test_domain_kernels_three_table_levels():
    """2^19 points: index bits 8..16 and 17..18 go through the two pre-summed table levels of the subset-sum kernels."""
    fc.check_domain_kernels(emu_lib.emu(), torch, CPU, "gf192", 19, 5, samples=(0, 255, 256, 65535, 65536, 131071, 131072, 300000))


@pytest.mark.parametrize("field_name,log_n,num_inputs", [("edwards_Fr", 7, 0), ("gf192", 6, 0), ("edwards_Fr", 6, 15)])
def test_native_indexer_and_prover_behind_the_c_abi(field_name, log_n, num_inputs):
    """iopx_fractal_index / iopx_fractal_prove (libiop_amd/cpp/fractal.hpp inside the library): roots and transcript equal the oracle's."""
    import oracle
    lib = emu_lib.emu()
    n = 1 << log_n
    inst = lib.aurora_example_instance({"gf192": 0, "edwards_Fr": 1}[field_name], n, num_inputs, n - 1, 0x2205)
    try:
        with pytest.raises(AssertionError):
            lib.fractal_prove(inst)                       # no index yet: std::logic_error
        roots = lib.fractal_index(inst)
        code = {"gf192": oracle.FIELD_GF192, "edwards_Fr": oracle.FIELD_EDWARDS}[field_name]
        ref, ref_roots = oracle.fractal_prove(code, log_n, num_inputs, 0x2205)
        assert roots == ref_roots
        assert lib.fractal_prove(inst) == ref
        assert lib.fractal_prove(inst) == ref
    finally:
        lib.aurora_instance_free(inst)
