"""Merkle layout of the oracle re-derived with hashlib from the layout rules of
libiop/bcs/merkle_tree.tcc:92-151,200-229 (what SURVEY.md §8 C1 reports as matching the reference's root)."""
import hashlib

import numpy as np
import pytest

import oracle
from helpers import rand_elems


def _py_tree(oracles, cs, additive, salts=None):
    n = oracles[0].shape[0]
    L = n // cs
    nodes = [None] * (2 * L - 1)
    for i in range(L):
        buf = b""
        for o in oracles:                       # oracle-major, then position within coset
            for j in range(cs):
                pos = i * cs + j if additive else i + j * L
                buf += o[pos].tobytes()
        d = hashlib.blake2b(buf, digest_size=32).digest()
        if salts is not None:
            d = hashlib.blake2b(d + salts[i].tobytes(), digest_size=32).digest()
        nodes[L - 1 + i] = d
    for j in range(L - 2, -1, -1):
        nodes[j] = hashlib.blake2b(nodes[2 * j + 1] + nodes[2 * j + 2], digest_size=32).digest()
    return nodes


@pytest.mark.parametrize("additive", [True, False])
@pytest.mark.parametrize("r,cs", [(1, 1), (1, 2), (2, 2), (4, 2), (1, 4), (4, 4), (12, 2)])
def test_tree_matches_hashlib(additive, r, cs):
    n = 16 * cs
    oracles = [rand_elems(100 + k, n, 3) for k in range(r)]
    nodes = oracle.merkle_build(oracles, cs, additive)
    exp = _py_tree(oracles, cs, additive)
    assert [bytes(x) for x in nodes] == exp


def test_zk_tree_with_fixed_salts():
    oracles = [rand_elems(5, 32, 3)]
    salts = np.random.default_rng(3).integers(0, 256, size=(16, 32), dtype=np.uint8)
    nodes = oracle.merkle_build(oracles, 2, True, salts)
    assert [bytes(x) for x in nodes] == _py_tree(oracles, 2, True, salts)


def test_rejects_bad_sizes():
    with pytest.raises(ValueError):             # merkle_tree.tcc:27-31
        oracle.merkle_build([rand_elems(1, 2, 3)], 2)
    with pytest.raises(ValueError):
        oracle.merkle_build([rand_elems(1, 12, 3)], 2)
