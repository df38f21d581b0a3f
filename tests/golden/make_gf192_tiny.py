#!/usr/bin/env python3
"""Writes tests/golden/gf192_tiny.json: one tiny GF(2^192) case of every step of the hot path, computed with PYTHON INTEGERS ONLY from the
definitions in the reference text — independent of oracle/'s C++ and of the kernels:

  * additive FFT over a 3-dimensional affine subspace = the polynomial's values at shift + sum_{bit k of i} basis[k], i = 0..7, in that order
    (libiop/algebra/utils.tcc:8-30 all_subset_sums; the reference's own test compares additive_FFT with this naive evaluation,
    libiop/tests/algebra/test_fft.cpp), and the inverse transform back to the coefficients;
  * one FRI fold with localization 1 and one with localization 2 (libiop/protocols/ldt/fri/fri_aux.tcc:36-103): the value at x_i of the
    polynomial of degree < 2^eta that interpolates f_i on each coset of 2^eta consecutive positions (Lagrange's formula), the property the
    reference's test checks (libiop/tests/protocols/test_fri.cpp);
  * a 4-leaf BLAKE2b Merkle tree over two oracles serialized by cosets of 2 (libiop/bcs/merkle_tree.tcc:92-151: slice[j + k * coset] = oracle k
    at coset position j, leaf = BLAKE2b-256 of the raw 24-byte elements, node = BLAKE2b-256(left || right), heap order), hashlib only.

Field: GF(2)[x] / (x^192 + x^7 + x^2 + x + 1), an element's bytes = its 192 coefficient bits little-endian (libff::gf192's three uint64
words).  This is the definition the repo restates from libff (absent from the tree): the vectors are independent of oracle/, not of that.

    python tests/golden/make_gf192_tiny.py        (rewrites the JSON next to it)
"""
import hashlib
import json
import os

P = (1 << 192) | 0x87
MASK = (1 << 192) - 1


def mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    for bit in range(r.bit_length() - 1, 191, -1):
        if (r >> bit) & 1:
            r ^= P << (bit - 192)
    return r


def power(a, e):
    r = 1
    while e:
        if e & 1:
            r = mul(r, a)
        a = mul(a, a)
        e >>= 1
    return r


def inv(a):
    return power(a, (1 << 192) - 2)


def element(basis, shift, i):
    e = shift
    for k, b in enumerate(basis):
        if (i >> k) & 1:
            e ^= b
    return e


def evaluate(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = mul(acc, x) ^ c
    return acc


def seeded(seed, count):
    """count field elements from SHA-256 in counter mode (any fixed, reproducible values will do)."""
    out = []
    for i in range(count):
        h = hashlib.sha256(b"gf192 tiny %d %d" % (seed, i)).digest()
        out.append(int.from_bytes(h[:24], "little"))
    return out


def lagrange_at(points, values, x):
    acc = 0
    for k, (xk, fk) in enumerate(zip(points, values)):
        num, den = 1, 1
        for l, xl in enumerate(points):
            if l != k:
                num = mul(num, x ^ xl)
                den = mul(den, xk ^ xl)
        acc ^= mul(fk, mul(num, inv(den)))
    return acc


def raw(e):
    return e.to_bytes(24, "little")


def h256(b):
    return hashlib.blake2b(b, digest_size=32).digest()


def main():
    m = 3
    basis = seeded(1, m)
    basis[0] |= 1                                        # three independent vectors (checked below)
    shift = seeded(2, 1)[0]
    pts = [element(basis, shift, i) for i in range(1 << m)]
    assert len(set(pts)) == 8, "basis vectors are dependent"
    coeffs = seeded(3, 8)
    evals = [evaluate(coeffs, x) for x in pts]
    short = coeffs[:3]                                   # a low-degree extension: 3 coefficients over the 8-point domain
    evals_short = [evaluate(short, x) for x in pts]

    folds = []
    for eta in (1, 2):
        x_i = seeded(10 + eta, 1)[0]
        cs = 1 << eta
        nxt = [lagrange_at(pts[j * cs:(j + 1) * cs], evals[j * cs:(j + 1) * cs], x_i) for j in range(8 // cs)]
        folds.append({"localization": eta, "x_i": hex(x_i), "next": [hex(v) for v in nxt]})

    other = seeded(4, 8)
    cs = 2
    leaves = [h256(b"".join(raw(v) for v in evals[j * cs:(j + 1) * cs]) + b"".join(raw(v) for v in other[j * cs:(j + 1) * cs])) for j in range(4)]
    n2 = [h256(leaves[0] + leaves[1]), h256(leaves[2] + leaves[3])]
    nodes = [h256(n2[0] + n2[1])] + n2 + leaves        # heap order: root, level 1, leaves

    out = {
        "field": "GF(2)[x] / (x^192 + x^7 + x^2 + x + 1); hex integers, bit i = coefficient of x^i",
        "basis": [hex(b) for b in basis], "shift": hex(shift),
        "coefficients": [hex(c) for c in coeffs], "evaluations": [hex(v) for v in evals],
        "short_coefficients": [hex(c) for c in short], "short_evaluations": [hex(v) for v in evals_short],
        "folds": folds,
        "merkle": {"oracles": [[hex(v) for v in evals], [hex(v) for v in other]], "coset_size": cs, "nodes": [n.hex() for n in nodes]},
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gf192_tiny.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
