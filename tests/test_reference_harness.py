"""SURVEY §8(b)'s last row / VERDICT r5 item 1b: the reference's OWN prover, compiled straight from /root/reference, (1) over a stand-in libff whose
field arithmetic is this repository's host code and (2) with the stubs of INTEGRATION.md compiled in verbatim, forwarding to the CPU build of the
kernel sources through the C ABI.  Both must produce the oracle's transcript byte for byte, and the reference's own verifier must accept it.

Container-only (skipped where /root/reference, libsodium or GMP are absent — the GPU box); tests/golden/reference_over_shim.json carries the digests there.
What this shows: libiop's protocol code calls the kernels UNCHANGED through the documented stubs, and the oracle follows libiop's logic as libiop's own code
executes it.  What it does not: anything about libff's bytes — the shim is a stand-in."""
import os
import sys

import pytest

import oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness"))
import harness  # noqa: E402

pytestmark = pytest.mark.skipif(harness.available() is not None, reason=str(harness.available()))

CODES = {"gf192": oracle.FIELD_GF192, "edwards_Fr": oracle.FIELD_EDWARDS}
SMALL = [c for c in harness.CASES if c[2] <= 10]


@pytest.fixture(scope="module")
def built():
    from emu_lib import emu
    emu()                       # tests/emu/libiopx_emu.so: the stubbed program links it
    harness.build()
    return True


_ORACLE = {}


def _oracle(protocol, field, log_n, k, seed, rs_extra):
    key = (protocol, field, log_n, k, seed, rs_extra)
    if key not in _ORACLE:
        if protocol == "aurora":
            _ORACLE[key] = (oracle.aurora_prove(CODES[field], log_n, k, seed, rs_extra=rs_extra), [])
        else:
            _ORACLE[key] = oracle.fractal_prove(CODES[field], log_n, k, seed, rs_extra=rs_extra)
    return _ORACLE[key]


@pytest.mark.parametrize("case", harness.CASES, ids=lambda c: "-".join(str(x) for x in c[:3]))
def test_the_references_own_prover_over_the_shim_equals_the_oracle(case, built):
    r = harness.run("plain", *case)
    t, roots = _oracle(*case)
    assert r["verifier_accepts"], "the reference's verifier rejected the reference's proof"
    assert r["transcript"] == t, "the oracle's transcript differs from the one libiop's own code produced"
    assert r["index_roots"] == roots
    assert r["kernel_launches_in_prover"] == {}


EXPECTED_KERNELS = {
    ("aurora", "gf192"): ("k_phase1_fwd", "k_phase1_inv", "k_rowcheck_add", "k_fz_add", "k_lincheck_add", "k_sumcheck_g_add_zero_sum", "k_fri_fold_fused_eta2", "k_ldt_combine_add_slots", "k_merkle_leaves_4x2", "k_merkle_level", "k_pow_blake2b"),
    ("aurora", "edwards_Fr"): ("k_mfft_pass", "k_rowcheck_fp", "k_fz_fp", "k_lincheck_fp", "k_sumcheck_g_fp", "k_fri_fold_fused_mul_eta2", "k_ldt_combine_fp", "k_merkle_leaves_4x2", "k_merkle_level", "k_pow_blake2b"),
    ("fractal", "gf192"): ("k_phase1_fwd", "k_phase1_inv", "k_ldt_combine_add_slots", "k_merkle_level", "k_pow_blake2b"),
    ("fractal", "edwards_Fr"): ("k_mfft_pass", "k_ldt_combine_fp", "k_merkle_level", "k_pow_blake2b"),
}


@pytest.mark.parametrize("case", SMALL, ids=lambda c: "-".join(str(x) for x in c[:3]))
def test_the_references_prover_calls_the_kernels_through_the_documented_stubs(case, built):
    """The stub text of INTEGRATION.md is the text compiled (tests/harness/make_shadow.py): transforms, folds, trees, the LDT combination and the
    proof of work of the reference's prover run in the kernel library — the launch counts say so — and nothing about the proof changes."""
    r = harness.run("stubbed", *case)
    t, roots = _oracle(*case)
    assert r["verifier_accepts"]
    assert r["transcript"] == t and r["index_roots"] == roots
    ran = r["kernel_launches_in_prover"]
    for k in EXPECTED_KERNELS[(case[0], case[1])]:
        assert ran.get(k, 0) > 0, (k, ran)


@pytest.mark.parametrize("field,log_n", [("gf192", 8), ("edwards_Fr", 9)])
def test_ligero_calls_the_kernels_unchanged_too(field, log_n, built):
    """north_star: "Aurora/Fractal/Ligero call it unchanged".  No oracle restates Ligero, so the two programs are compared with each other: the reference's
    Ligero prover (instrument_ligero_snark.cpp's settings, non-zk) with the stubs compiled in produces the transcript the plain one does, its transforms,
    tree, combination and proof of work having run in the kernel library, and the reference's verifier accepts."""
    plain = harness.run("plain", "ligero", field, log_n, 15, 0x2206, 2)
    stubbed = harness.run("stubbed", "ligero", field, log_n, 15, 0x2206, 2)
    assert plain["verifier_accepts"] and stubbed["verifier_accepts"]
    assert plain["transcript"] == stubbed["transcript"] and len(plain["transcript"]) > 10000
    ran = stubbed["kernel_launches_in_prover"]
    for k in (("k_phase1_fwd", "k_phase1_inv") if field == "gf192" else ("k_mfft_pass",)) + ("k_merkle_leaves", "k_pow_blake2b"):
        assert ran.get(k, 0) > 0, (k, ran)
    assert plain["kernel_launches_in_prover"] == {}


def test_the_compiled_stub_text_is_integration_md(built):
    with open(os.path.join(harness.ROOT, "INTEGRATION.md")) as f:
        md = f.read()
    with open(os.path.join(harness.HERE, "_build", "stubs.inc")) as f:
        inc = f.read()
    for needle in ("additive_FFT<libff::gf192>", "multiplicative_IFFT<libff::edwards_Fr>", "construct_with_leaves_serialized_by_cosets", "solve_pow_internal",
                   "combined_LDT_virtual_oracle<libff::edwards_Fr>::evaluated_contents"):
        assert needle in inc and needle in md
    body = [l for l in inc.splitlines() if l.strip() and not l.startswith("// generated") and l not in ("namespace libiop {", "} // namespace libiop")]
    missing = [l for l in body if l not in md]
    assert not missing, missing[:3]


def test_the_committed_digests_are_what_the_reference_program_produces(built):
    import json
    with open(os.path.join(harness.ROOT, "tests", "golden", "reference_over_shim.json")) as f:
        entries = json.load(f)["entries"]
    assert len(entries) == len(harness.CASES)
    for e in entries[:3] + entries[7:8]:
        r = harness.run("plain", e["protocol"], e["field"], e["log_n"], e["num_inputs"], e["seed"], e["rs_extra"])
        assert harness.digest(r["transcript"]) == e["transcript_blake2b"] and len(r["transcript"]) == e["transcript_bytes"]


def test_the_committed_function_vectors_are_what_the_references_functions_produce(built):
    """tests/golden/reference_functions.json == the lines tests/harness/reference_vectors.cpp prints today (libiop's own FFTs, folds, trees, ... on the seeds)."""
    import json
    import subprocess
    harness._make(["_build/reference_vectors"])
    out = subprocess.run([os.path.join(harness.HERE, "_build", "reference_vectors")], capture_output=True, text=True, check=True).stdout
    now = [json.loads(line) for line in out.splitlines() if line.startswith("{")]
    with open(os.path.join(harness.ROOT, "tests", "golden", "reference_functions.json")) as f:
        committed = json.load(f)["entries"]
    assert now == committed and len(now) > 500


@pytest.mark.parametrize("name", harness.reftests())
def test_the_references_own_test_files_pass_with_the_kernels_underneath(name, built):
    """libiop's own test files (libiop/tests/*/test_*.cpp), unmodified, compiled as programs with the stubs of INTEGRATION.md in force (googletest replaced by
    tests/harness/shim/gtest/gtest.h, tests_prelude.hpp force-included ahead of the test's text): every TEST passes, and the multiplicative ones ran their
    transforms in the kernel library.  Default: three programs; IOPX_REFTESTS=all: the 34 of tests/harness/Makefile (143 tests with the stubs, 145 without, recorded in
    profiles/r06_reference_own_tests_{plain,stubbed}.json)."""
    harness.build_reftest(name)
    ran, ok, kernels, tail = harness.run_reftest(name)
    assert ran > 0 and ok == ran, (name, ran, ok, tail)
    if name in ("algebra/test_fft", "protocols/test_aurora_protocol", "protocols/test_direct_ldt"):
        assert kernels.get("k_mfft_pass", 0) > 0, kernels
    if name == "protocols/test_fri_aux":
        assert kernels.get("k_fri_fold_fused_mul_eta2", 0) > 0, kernels
