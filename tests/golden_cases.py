"""tests/golden/gf192_tiny.json (pure-Python integers, tests/golden/make_gf192_tiny.py) applied to an implementation of the hot path:
`impl` is the ctypes binding of the product library (GPU), of its CPU emulation build, or an adapter around the oracle."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "gf192_tiny.json")) as f:
        return json.load(f)


def words(hex_list):
    out = np.zeros((len(hex_list), 3), dtype=np.uint64)
    for i, h in enumerate(hex_list):
        v = int(h, 16)
        for w in range(3):
            out[i, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def check(fft, ifft, fold, merkle_nodes):
    """fft(coeffs, basis, shift) / ifft(evals, basis, shift) / fold(f, basis, shift, coset_size, x) -> (n, 3) uint64 arrays;
    merkle_nodes(oracles, coset_size) -> (2L - 1, 32) uint8."""
    g = load()
    basis, shift = words(g["basis"]), words([g["shift"]])[0]
    coeffs, evals = words(g["coefficients"]), words(g["evaluations"])
    assert np.array_equal(fft(coeffs, basis, shift), evals), "additive FFT differs from the integer evaluation"
    assert np.array_equal(ifft(evals, basis, shift), coeffs), "additive IFFT differs"
    assert np.array_equal(fft(words(g["short_coefficients"]), basis, shift), words(g["short_evaluations"])), "low-degree extension differs"
    for case in g["folds"]:
        got = fold(evals, basis, shift, 1 << case["localization"], words([case["x_i"]])[0])
        assert np.array_equal(got, words(case["next"])), "FRI fold (localization %d) differs from Lagrange interpolation" % case["localization"]
    mk = g["merkle"]
    nodes = merkle_nodes([words(o) for o in mk["oracles"]], mk["coset_size"])
    assert [bytes(n).hex() for n in nodes] == mk["nodes"], "Merkle nodes differ from hashlib's"
