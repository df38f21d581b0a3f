"""tests/golden/edwards_tiny.json (pure-Python integers, tests/golden/make_edwards_tiny.py) applied to an implementation of the prime-field arm of
the hot path: the ctypes binding of the product library (GPU), of its CPU emulation build, or an adapter around the oracle."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P = 1552511030102430251236801561344621993261920897571225601


def load():
    with open(os.path.join(HERE, "golden", "edwards_tiny.json")) as f:
        return json.load(f)


def mont_words(hex_list):
    """canonical hex integers -> (count, 3) uint64 Montgomery words (x * 2^192 mod p), computed here with Python integers"""
    out = np.zeros((len(hex_list), 3), dtype=np.uint64)
    for i, h in enumerate(hex_list):
        v = int(h, 16) * (1 << 192) % P
        for w in range(3):
            out[i, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def check(fft, ifft, ifft_known_degree, fold, merkle_nodes):
    """fft(coeffs, log_n, shift) / ifft(evals, shift) / ifft_known_degree(evals, degree, shift) / fold(f, shift, coset_size, x) -> (n, 3) uint64 Montgomery
    words; merkle_nodes(oracles, coset_size) -> (2L - 1, 32) uint8 with the multiplicative position map."""
    g = load()
    assert int(g["p"], 16) == P
    shift, one = mont_words([g["shift"]])[0], mont_words(["0x1"])[0]
    coeffs, evals = mont_words(g["coefficients"]), mont_words(g["evaluations"])
    assert np.array_equal(fft(coeffs, 3, shift), evals), "multiplicative FFT differs from the integer evaluation"
    assert np.array_equal(fft(coeffs, 3, one), mont_words(g["unshifted_evaluations"])), "FFT over the plain subgroup differs"
    assert np.array_equal(ifft(evals, shift), coeffs), "multiplicative IFFT differs"
    assert np.array_equal(fft(mont_words(g["short_coefficients"]), 3, shift), mont_words(g["short_evaluations"])), "degree-aware FFT (3 coefficients) differs"
    got = ifft_known_degree(mont_words(g["four_evaluations"]), 4, shift)
    assert np.array_equal(got, mont_words(g["four_coefficients"])), "IFFT of known degree differs"
    for case in g["folds"]:
        got = fold(evals, shift, 1 << case["localization"], mont_words([case["x_i"]])[0])
        assert np.array_equal(got, mont_words(case["next"])), "FRI fold (localization %d) differs from Lagrange interpolation" % case["localization"]
    mk = g["merkle"]
    nodes = merkle_nodes([mont_words(o) for o in mk["oracles"]], mk["coset_size"])
    assert [bytes(x).hex() for x in nodes] == mk["nodes"], "Merkle nodes differ from hashlib's"
