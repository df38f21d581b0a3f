"""tests/golden/reference_over_shim.json: transcripts of libiop's own prover (tests/harness, generated in the build container by
tests/golden/make_reference_over_shim.py) — shared by the oracle, CPU-kernel and HIP tests."""
import hashlib
import json
import os

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODES = {"gf192": oracle.FIELD_GF192, "edwards_Fr": oracle.FIELD_EDWARDS}
NAMES = {"gf192": "gf192", "edwards_Fr": "edwards_Fr"}


def entries(large_up_to=0):
    """The small cases (2^6 - 2^12), plus the "large_entries" up to 2^large_up_to (2^14: about 25 s of the oracle prover)."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_over_shim.json")) as f:
        doc = json.load(f)
    return doc["entries"] + [e for e in doc["large_entries"] if e["log_n"] <= large_up_to]


def ident(e):
    return "%s-%s-%d" % (e["protocol"], e["field"], e["log_n"])


def check_bytes(e, transcript, roots):
    assert e["reference_verifier_accepts"]
    assert len(transcript) == e["transcript_bytes"]
    assert hashlib.blake2b(transcript, digest_size=32).hexdigest() == e["transcript_blake2b"], "transcript differs from the one libiop's own prover produced"
    assert [bytes(r).hex() for r in roots] == e["index_roots"]


def check_oracle(e):
    if e["protocol"] == "aurora":
        t, roots = oracle.aurora_prove(CODES[e["field"]], e["log_n"], e["num_inputs"], e["seed"], rs_extra=e["rs_extra"], localization=e["localization"]), []
    else:
        t, roots = oracle.fractal_prove(CODES[e["field"]], e["log_n"], e["num_inputs"], e["seed"], rs_extra=e["rs_extra"], localization=e["localization"])
    check_bytes(e, t, roots)


def check_native(lib, e):
    """The native prover behind the C ABI (default parameters = the instrument programs': the fixture's)."""
    import head_cases as hc
    t, roots, _ = hc.prove(lib, e["protocol"], NAMES[e["field"]], e["log_n"], e["num_inputs"], e["seed"])
    check_bytes(e, t, roots)
