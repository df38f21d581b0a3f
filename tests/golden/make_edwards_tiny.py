#!/usr/bin/env python3
"""Writes tests/golden/edwards_tiny.json: one tiny case of every step of the PRIME-FIELD arm of the hot path (BASELINE configs[0] and configs[4]:
the 181-bit scalar field of the Edwards curve, multiplicative cosets), computed with PYTHON INTEGERS ONLY from the definitions in the
reference text — independent of oracle/'s C++ and of the kernels:

  * multiplicative FFT over the coset shift * <g> of order 8 = the polynomial's values at shift * g^i, i = 0..7, in that order
    (libiop/algebra/field_subset/subgroup.tcc:55-59: g = multiplicative_generator^((p - 1) / 8); the reference's own test compares
    multiplicative_FFT with naive evaluation, libiop/tests/algebra/test_fft.cpp), with fewer coefficients than points and with a
    NON-power-of-two coefficient count (the degree-aware branch, fft.tcc:236-317), and the inverse transform back to the coefficients;
  * IFFT_of_known_degree_over_field_subset (fft.tcc:435-456): the coefficients from every second evaluation;
  * one FRI fold with localization 1 and one with localization 2 (libiop/protocols/ldt/fri/fri_aux.tcc:106-249): the value at x_i of the
    polynomial of degree < 2^eta interpolating f_i on each coset {j + k n / 2^eta} (subgroup.tcc:175-197), by Lagrange's formula;
  * a 4-leaf BLAKE2b Merkle tree over two oracles serialized by multiplicative cosets of 2 (merkle_tree.tcc:92-151 with the position map of
    subgroup.tcc:175-197: leaf j holds positions j and j + 4 of every oracle), hashlib only.

Field: integers mod p = 1552511030102430251236801561344621993261920897571225601 (181 bits, 2-adicity 31), multiplicative_generator 19.  An
element's 24 bytes are libff's Fp_model `mont_repr`: (x * 2^192 mod p) as three little-endian uint64 words.  These are the facts the repo
restates from libff (absent from the tree): the vectors are independent of oracle/, not of them.

    python tests/golden/make_edwards_tiny.py        (rewrites the JSON next to it)
"""
import hashlib
import json
import os

P = 1552511030102430251236801561344621993261920897571225601
GENERATOR = 19
R = 1 << 192


def inv(a):
    return pow(a, P - 2, P)


def evaluate(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def seeded(seed, count):
    out = []
    for i in range(count):
        h = hashlib.sha256(b"edwards tiny %d %d" % (seed, i)).digest()
        out.append(int.from_bytes(h, "little") % P)
    return out


def lagrange_at(points, values, x):
    acc = 0
    for k, (xk, fk) in enumerate(zip(points, values)):
        num, den = 1, 1
        for l, xl in enumerate(points):
            if l != k:
                num = num * (x - xl) % P
                den = den * (xk - xl) % P
        acc = (acc + fk * num % P * inv(den)) % P
    return acc


def mont(x):
    return x * R % P


def raw(x):
    return mont(x).to_bytes(24, "little")


def h256(b):
    return hashlib.blake2b(b, digest_size=32).digest()


def main():
    n = 8
    assert (P - 1) % (1 << 31) == 0 and (P - 1) % (1 << 32) != 0
    g = pow(GENERATOR, (P - 1) // n, P)
    assert pow(g, n, P) == 1 and pow(g, n // 2, P) != 1
    shift = GENERATOR                                    # the codeword shift of the reference's provers (subgroup.tcc:311-315)
    pts = [shift * pow(g, i, P) % P for i in range(n)]
    coeffs = seeded(3, 8)
    evals = [evaluate(coeffs, x) for x in pts]
    short = coeffs[:3]                                   # non-power-of-two coefficient count
    evals_short = [evaluate(short, x) for x in pts]
    four = coeffs[:4]
    evals_four = [evaluate(four, x) for x in pts]        # IFFT_of_known_degree(evals_four, 4, domain) reads positions 0, 2, 4, 6
    unshifted = [evaluate(coeffs, pow(g, i, P)) for i in range(n)]   # shift 1: the plain subgroup

    folds = []
    for eta in (1, 2):
        x_i = seeded(10 + eta, 1)[0]
        cs = 1 << eta
        q = n // cs
        nxt = [lagrange_at([pts[j + k * q] for k in range(cs)], [evals[j + k * q] for k in range(cs)], x_i) for j in range(q)]
        folds.append({"localization": eta, "x_i": hex(x_i), "next": [hex(v) for v in nxt]})

    other = seeded(4, 8)
    cs, L = 2, 4
    leaves = [h256(b"".join(raw(evals[j + k * L]) for k in range(cs)) + b"".join(raw(other[j + k * L]) for k in range(cs))) for j in range(L)]
    n2 = [h256(leaves[0] + leaves[1]), h256(leaves[2] + leaves[3])]
    nodes = [h256(n2[0] + n2[1])] + n2 + leaves          # heap order: root, level 1, leaves

    out = {
        "field": "integers mod p (canonical values as hex); device bytes = (x * 2^192 mod p) as three little-endian uint64 words",
        "p": hex(P), "generator": hex(g), "shift": hex(shift),
        "coefficients": [hex(c) for c in coeffs], "evaluations": [hex(v) for v in evals],
        "short_coefficients": [hex(c) for c in short], "short_evaluations": [hex(v) for v in evals_short],
        "four_coefficients": [hex(c) for c in four], "four_evaluations": [hex(v) for v in evals_four],
        "unshifted_evaluations": [hex(v) for v in unshifted],
        "folds": folds,
        "merkle": {"oracles": [[hex(v) for v in evals], [hex(v) for v in other]], "coset_size": cs, "nodes": [x.hex() for x in nodes]},
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "edwards_tiny.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
