"""General constraint systems through iopx_aurora_instance_create on the MI355X, 2^8 - 2^12 constraints, both fields, native and Python provers,
byte-equal to the oracle provers fed the same CSR triples; the unsatisfied variants' bytes are the oracle's too (tests/general_cases.py; the
CPU-compiled run is tests/test_general_r1cs_emu.py)."""
import pytest
import torch

import general_cases as gc

pytestmark = pytest.mark.gpu
BOTH = ["gf192", "edwards_Fr"]


@pytest.fixture(scope="module")
def gpu():
    import libiop_amd
    lib = libiop_amd.lib()
    lib.init(0)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)      # DeviceOps (the second prover) shares torch's stream
    return lib


DEV = torch.device("cuda:0")


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("num_constraints,num_variables", [(4096, 4095), (1024, 8191)])
def test_spmv_against_the_oracle(gpu, field_name, num_constraints, num_variables):
    gc.check_spmv(gpu, torch, DEV, field_name, num_constraints, num_variables, 6)


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("num_constraints,num_variables,num_inputs", [(256, 255, 15), (4096, 4095, 15), (1024, 4095, 31), (2048, 511, 7), (512, 511, 0)])
def test_aurora_on_general_instances(gpu, field_name, num_constraints, num_variables, num_inputs, monkeypatch):
    gc.check_aurora(gpu, torch, DEV, monkeypatch, field_name, num_constraints, num_variables, num_inputs, 50 + num_inputs)


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("kind", ["constraint", "primary", "auxiliary"])
@pytest.mark.parametrize("log_n", [8, 11])
def test_aurora_unsatisfied_bytes_are_the_oracles(gpu, field_name, kind, log_n, monkeypatch):
    n = 1 << log_n
    gc.check_aurora_unsatisfied(gpu, torch, DEV, monkeypatch, field_name, n, n - 1, 15, 60 + log_n, kind)


@pytest.mark.parametrize("field_name,num_constraints,num_inputs,max_nnz", [("gf192", 256, 15, None), ("gf192", 1024, 15, 700), ("edwards_Fr", 4096, 15, None),
                                                                           ("edwards_Fr", 2048, 0, None), ("edwards_Fr", 1024, 31, 400)])
def test_fractal_on_general_instances(gpu, field_name, num_constraints, num_inputs, max_nnz, monkeypatch):
    gc.check_fractal(gpu, torch, DEV, monkeypatch, field_name, num_constraints, num_inputs, 70 + num_inputs, max_nnz=max_nnz)


@pytest.mark.parametrize("field_name,num_constraints", [("gf192", 256), ("edwards_Fr", 2048)])
@pytest.mark.parametrize("kind", ["constraint", "primary", "auxiliary"])
def test_fractal_unsatisfied_bytes_are_the_oracles(gpu, field_name, num_constraints, kind, monkeypatch):
    gc.check_fractal(gpu, torch, DEV, monkeypatch, field_name, num_constraints, 15, 80, kind=kind)


def test_aurora_general_instance_at_2p13_equals_the_oracle_prover(gpu, monkeypatch):
    """A larger general instance against the oracle prover itself: byte-equality at 2^13 constraints over GF(2^192) (2^16 goes against the recorded digest below)."""
    gc.check_aurora(gpu, torch, DEV, monkeypatch, "gf192", 1 << 13, (1 << 13) - 1, 15, 91)


def test_aurora_general_instance_at_2p16_native_equals_python_and_the_oracle_verifier_accepts(gpu, monkeypatch):
    """Beyond the oracle PROVER's reach in a test (minutes): at 2^16 the two independently written device provers must agree byte for byte and the
    oracle's VERIFIER (point evaluations only) must accept the proof of the general instance; a flipped answer is rejected."""
    import oracle
    import r1cs_general as rg
    from libiop_amd import aurora, domains
    n, k = 1 << 16, 15
    inst = rg.generate("gf192", n, n - 1, k, 92)
    code = oracle.FIELD_GF192
    t, prof = gc.native_aurora(gpu, 0, inst, monkeypatch, True)
    ops = domains.DeviceOps(gpu, torch, DEV, domains.GF192())
    cs, primary, auxiliary = gc.python_cs(ops, inst)
    params = aurora.AuroraParameters(ops.field, n, n - 1, k)
    assert aurora.aurora_snark_prover(ops, cs, primary, auxiliary, params).serialize() == t
    assert oracle.aurora_verify_csr(code, inst.matrices, n - 1, k, inst.assignment[:k], t)
    bad = bytearray(t)
    bad[len(bad) // 2] ^= 1
    assert not oracle.aurora_verify_csr(code, inst.matrices, n - 1, k, inst.assignment[:k], bytes(bad))


def _recorded_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_general_r1cs_digests.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_general_instances_against_recorded_oracle_digests(gpu, index, monkeypatch):
    """Sizes the oracle provers need minutes for (2^16 Aurora over both fields, 2^15 Fractal over the 181-bit field): their transcripts' digests and index
    roots are a fixture (tools/make_general_digests.py ran the oracle on the CPU); the native provers' bytes must hash to them."""
    import hashlib
    import r1cs_general as rg
    case = _recorded_cases()[index]
    n, k = case["num_constraints"], case["num_inputs"]
    inst = rg.generate(case["field"], n, n - 1, k, case["seed"], max_nnz=case["max_nnz"])
    assert inst.nnz() == case["nnz"]                                  # the generator still builds the recorded instance
    native_code = gc.FIELDS[case["field"]][1]
    if case["protocol"] == "aurora":
        t, _ = gc.native_aurora(gpu, native_code, inst, monkeypatch, True)
        roots = []
    else:
        t, roots = gc.native_fractal(gpu, native_code, inst, monkeypatch, True)
    assert [r.hex() for r in roots] == case["index_roots"]
    assert len(t) == case["argument_bytes"] and hashlib.blake2b(t, digest_size=32).hexdigest() == case["transcript_blake2b"]
