"""The REFERENCE'S OWN provers on the MI355X: libiop's sources, compiled unmodified in the build container with the stubs of INTEGRATION.md compiled in
(tests/harness, `make hip`), linked against the HIP library.  The binary travels with the snapshot (tests/harness/_hip/, git-ignored); where it is absent
— a tree that was never built next to /root/reference — the test is skipped.  Nothing here reads /root/reference."""
import hashlib
import json
import os
import subprocess

import pytest

import reference_digest_cases as rc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "harness", "_hip", "reference_stubbed_hip")

EXPECTED_KERNELS = {("aurora", "gf192"): ("k_bfly_edge_fwd", "k_rowcheck_add", "k_fz_add", "k_lincheck_add", "k_sumcheck_g_add_zero_sum", "k_fri_fold_fused_eta2", "k_ldt_combine_add_slots", "k_merkle_level", "k_pow_blake2b"),
                    ("aurora", "edwards_Fr"): ("k_mfft_pass", "k_rowcheck_fp", "k_fz_fp", "k_lincheck_fp", "k_sumcheck_g_fp", "k_fri_fold_fused_mul_eta2", "k_ldt_combine_fp", "k_merkle_level", "k_pow_blake2b"),
                    ("fractal", "gf192"): ("k_bfly_edge_fwd", "k_ldt_combine_add_slots", "k_merkle_level", "k_pow_blake2b"),
                    ("fractal", "edwards_Fr"): ("k_mfft_pass", "k_ldt_combine_fp", "k_merkle_level", "k_pow_blake2b")}


@pytest.mark.skipif(not os.path.exists(EXE), reason="tests/harness/_hip/reference_stubbed_hip was not built (needs /root/reference at build time)")
@pytest.mark.parametrize("e", [e for e in rc.entries() if e["log_n"] >= 10], ids=rc.ident)
def test_the_references_own_prover_runs_on_the_hip_kernels(e, tmp_path):
    out = str(tmp_path / "t.bin")
    r = subprocess.run([EXE, e["protocol"], e["field"], str(e["log_n"]), str(e["num_inputs"]), hex(e["seed"]), str(e["rs_extra"]), str(e["localization"]), out],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-800:], r.stderr[-800:])
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["verifier_accepts"], "the reference's verifier rejected the proof its own prover produced on the HIP kernels"
    with open(out, "rb") as f:
        t = f.read()
    assert len(t) == e["transcript_bytes"] and hashlib.blake2b(t, digest_size=32).hexdigest() == e["transcript_blake2b"]
    assert info["index_roots"] == e["index_roots"]
    for k in EXPECTED_KERNELS[(e["protocol"], e["field"])]:
        assert info["kernel_launches_in_prover"].get(k, 0) > 0, (k, info["kernel_launches_in_prover"])
