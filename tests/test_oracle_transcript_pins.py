"""The oracle provers' transcripts for small seeded instances against the digests committed in tests/golden/transcript_digests.json
(made by tools/make_transcript_digests.py): a pin of the oracle across rounds.  The oracle's verifier accepts each of them."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import make_transcript_digests as mk


def test_oracle_transcripts_match_committed_digests():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transcript_digests.json")) as f:
        committed = json.load(f)
    assert mk.digests() == committed


@pytest.mark.parametrize("case", mk.CASES["fractal"][1:], ids=lambda c: "%s-2^%d-k%d" % (c[0], c[2], c[3]))
def test_oracle_fractal_self_consistency(case):
    import oracle
    _, code, log_n, k, seed = case
    t, roots = oracle.fractal_prove(code, log_n, k, seed)
    assert oracle.fractal_verify(code, log_n, k, seed, t, roots)
    bad = bytearray(t); bad[len(bad) // 3] ^= 4
    assert not oracle.fractal_verify(code, log_n, k, seed, bytes(bad), roots)
    if k:
        assert not oracle.fractal_verify(code, log_n, k, seed + 1, t, roots)       # another statement (primary input)
