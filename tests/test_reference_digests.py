"""The oracle and the native provers (CPU build of the kernel sources) against transcripts of libiop's OWN prover: tests/golden/reference_over_shim.json,
generated in the build container by tests/harness (the reference's sources compiled unmodified over a stand-in libff).  Runs everywhere — the fixture is data."""
import pytest

import emu_lib
import reference_digest_cases as rc


# every small entry but the 2^12 Aurora one over GF(2^192) (20 s of the oracle prover; tests/test_reference_harness.py runs it where the reference tree exists, and the
# oracle's equality at 2^16 / 2^18 / 2^20 is checked digest against digest below)
@pytest.mark.parametrize("e", [e for e in rc.entries() if not (e["field"] == "gf192" and e["log_n"] == 12)], ids=rc.ident)
def test_oracle_equals_the_references_own_prover(e):
    rc.check_oracle(e)


@pytest.mark.parametrize("e", [e for e in rc.entries() if e["log_n"] <= 8], ids=rc.ident)
def test_native_prover_on_cpu_kernels_equals_the_references_own_prover(e):
    rc.check_native(emu_lib.emu(), e)


def test_the_references_own_prover_at_baseline_sizes_equals_the_oracle_goldens():
    """"large_entries": libiop's own prover (tests/harness, one core of the build container, minutes to an hour and up to 33 GB per run) at 2^16 – 2^20.  The
    oracle prover's digests at those sizes are the fixtures the HIP provers are compared with on the MI355X (tests/test_gpu_fullsize.py; bench.py prints the
    timed proof's): they are the same digests, so the timed 2^20 proofs are byte for byte what libiop's code produces."""
    import json
    import os
    large = {(e["protocol"], e["field"], e["log_n"]): e for e in __import__("json").load(open(os.path.join(rc.ROOT, "tests", "golden", "reference_over_shim.json")))["large_entries"]}
    assert ("fractal", "edwards_Fr", 20) in large and ("aurora", "gf192", 20) in large        # BASELINE configs[4] and configs[3] (the headline), one process
    with open(os.path.join(rc.ROOT, "tests", "golden", "oracle_aurora_transcript_digests_large.json")) as f:
        aurora = json.load(f)["digests"]
    with open(os.path.join(rc.ROOT, "tests", "golden", "oracle_fractal_transcript_digests_large.json")) as f:
        fractal = json.load(f)["digests"]
    for (protocol, field, log_n), e in large.items():
        assert e["reference_verifier_accepts"]
        if log_n < 16:
            continue                        # 2^14: the oracle prover itself is run against it above
        gold = (aurora if protocol == "aurora" else fractal)[str(log_n)]
        assert e["transcript_blake2b"] == gold["transcript_blake2b"], (protocol, log_n)
        if protocol == "fractal":
            assert e["index_roots"] == gold["index_roots"] and e["transcript_bytes"] == gold["argument_bytes"]
