"""General constraint systems through iopx_aurora_instance_create — the only way a real caller reaches the native provers — against the
oracle provers fed the same CSR triples (tests/general_cases.py, tests/r1cs_general.py).  Kernel sources compiled for the CPU (tests/emu);
the same cases at 2^8 - 2^12 run on the MI355X in tests/test_gpu_general_r1cs.py."""
import pytest
import torch

import emu_lib
import general_cases as gc
import oracle
import r1cs_general as rg

CPU = torch.device("cpu")
BOTH = ["gf192", "edwards_Fr"]


@pytest.mark.parametrize("field_name", BOTH)
def test_generator_instances_are_what_they_claim(field_name):
    """Shape of the generated systems (so that the cases below really cover multi-term rows, constant-column terms, empty rows, repeated and hot
    columns, non-unit coefficients) and the oracle's is_satisfied on them and on the three perturbations."""
    code = gc.FIELDS[field_name][0]
    inst = rg.generate(field_name, 256, 255, 15, 11)
    A, B, C = inst.rows
    for M in (A, B):
        assert any(len(r) == 0 for r in M) and any(len(r) >= 4 for r in M)
        assert any(c == 0 for r in M for c, _ in r)                                          # the constant 1
        assert any(len({c for c, _ in r}) < len(r) for r in M)                               # one variable twice in a row
        assert sum(v != inst.F.one for r in M for _, v in r) > len(M)                        # coefficients != 1
        counts = {}
        for r in M:
            for c, _ in r:
                counts[c] = counts.get(c, 0) + 1
        assert max(counts.values()) >= 20                                                    # a column hit by many rows
    assert oracle.r1cs_check_csr(code, inst.matrices, 255, 15, inst.assignment)[0] == 0
    assert oracle.r1cs_check_csr(code, rg.perturbed(inst, "constraint", 1).matrices, 255, 15, inst.assignment)[0] == 1
    for kind in ("primary", "auxiliary"):
        bad = rg.perturbed(inst, kind, 1)
        assert oracle.r1cs_check_csr(code, bad.matrices, 255, 15, bad.assignment)[0] >= 1
    sparse = rg.generate(field_name, 128, 127, 15, 12, max_nnz=128)
    assert max(sparse.nnz()) <= 128 and min(sparse.nnz()) > 64 and any(len(r) >= 2 for r in sparse.rows[0])


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("num_constraints,num_variables", [(128, 127), (64, 255)])
def test_spmv_against_the_oracle(field_name, num_constraints, num_variables):
    gc.check_spmv(emu_lib.emu(), torch, CPU, field_name, num_constraints, num_variables, 5)


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("num_constraints,num_variables,num_inputs", [(256, 255, 15), (128, 511, 15), (512, 127, 7), (128, 127, 0)])
def test_aurora_on_general_instances(field_name, num_constraints, num_variables, num_inputs, monkeypatch):
    gc.check_aurora(emu_lib.emu(), torch, CPU, monkeypatch, field_name, num_constraints, num_variables, num_inputs, 30 + num_inputs)


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("kind", ["constraint", "primary", "auxiliary"])
def test_aurora_unsatisfied_bytes_are_the_oracles(field_name, kind, monkeypatch):
    gc.check_aurora_unsatisfied(emu_lib.emu(), torch, CPU, monkeypatch, field_name, 256, 255, 15, 21, kind)


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("num_inputs,max_nnz", [(15, None), (15, 60), (0, None)])
def test_fractal_on_general_instances(field_name, num_inputs, max_nnz, monkeypatch):
    gc.check_fractal(emu_lib.emu(), torch, CPU, monkeypatch, field_name, 128, num_inputs, 41 + num_inputs, max_nnz=max_nnz)


@pytest.mark.parametrize("field_name", BOTH)
@pytest.mark.parametrize("kind", ["constraint", "primary", "auxiliary"])
def test_fractal_unsatisfied_bytes_are_the_oracles(field_name, kind, monkeypatch):
    gc.check_fractal(emu_lib.emu(), torch, CPU, monkeypatch, field_name, 128, 15, 44, kind=kind, python_prover=(kind == "constraint"))


def test_recorded_general_instances_are_reproducible():
    """tests/golden/oracle_general_r1cs_digests.json (oracle transcripts of larger general instances, compared on the GPU): the seeded generator must still build
    the recorded instances — same entry counts per matrix — or the fixture would silently stop meaning anything."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_general_r1cs_digests.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 4
    for case in cases:
        if case["num_constraints"] > (1 << 12):
            continue                                                   # the large ones take the generator tens of seconds: checked by the GPU test
        n = case["num_constraints"]
        inst = rg.generate(case["field"], n, n - 1, case["num_inputs"], case["seed"], max_nnz=case["max_nnz"])
        assert inst.nnz() == case["nnz"]
