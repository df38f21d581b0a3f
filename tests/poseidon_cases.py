"""Poseidon parity cases shared by the CPU-emulation suite and the GPU suite: each takes the loaded C-ABI library
(`lib`) and compares it with the oracle / the reference's known answers (tests/golden/poseidon_kat.json)."""
import json
import os

import numpy as np

import libiop_amd
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon_kat.json")))
SETS = json.load(open(os.path.join(ROOT, "libiop_amd", "data", "poseidon_alt_bn128.json")))["sets"]
SET_NAMES = ["test_params", "starkware_alpha5_t3", "high_alpha17_t3", "high_alpha17_t4"]


def param_pair(name):
    d = KAT["test_params"] if name == "test_params" else SETS[name]
    return libiop_amd.PoseidonParams.from_dict(d), oracle.PoseidonParams(d)


def rand_bn(seed, count):
    """Montgomery words of uniformly random elements (plus the edge values 0, 1, p - 1 at the front)."""
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(40), "little") % oracle.BN128_R for _ in range(count)]
    for i, v in enumerate([0, 1, oracle.BN128_R - 1][:count]):
        vals[i] = v
    return oracle.bn_from_ints(vals)


def check_to_montgomery(lib):
    xs = [0, 1, 5, oracle.BN128_R - 1, oracle.BN128_R, oracle.BN128_R + 7, (1 << 256) - 1, 12345678901234567890123456789]
    assert np.array_equal(lib.bn128_to_montgomery(xs), oracle.bn_from_ints([x % oracle.BN128_R for x in xs]))


def check_permutation_kats(lib):
    # test_poseidon.cpp:46-55 and :61-65: element 0 of permutation(zero state)
    p, _ = param_pair("test_params")
    st = lib.poseidon_permute(p, np.zeros((1, 3, 4), dtype=np.uint64))
    assert oracle.bn_to_ints(st[0, :1])[0] == KAT["zero_state_squeeze_test_params"]
    p, _ = param_pair("high_alpha17_t3")
    st = lib.poseidon_permute(p, np.zeros((1, 3, 4), dtype=np.uint64))
    assert oracle.bn_to_ints(st[0, :1])[0] == KAT["zero_state_squeeze_high_alpha_t3"]


def check_permutation(lib, name, count=70):
    p, po = param_pair(name)
    t = p.state_size
    st = rand_bn(11, count * t).reshape(count, t, 4)
    got = lib.poseidon_permute(p, st)
    for i in range(count):
        assert np.array_equal(got[i], oracle.poseidon_permute(po, st[i])), (name, i)


def check_merkle(lib, name, r, cs, L, additive, zk):
    p, po = param_pair(name)
    n = L * cs
    oracles = [rand_bn(100 + k, n) for k in range(r)]
    salts = np.random.default_rng(9).integers(0, 256, size=(L, 32), dtype=np.uint8) if zk else None
    if zk:
        salts[0, :8] = 0xFF         # a salt whose integer exceeds the modulus (first word is the most significant)
    got = lib.merkle_tree_poseidon(p, oracles, cs, 0 if additive else 1, salts)
    assert np.array_equal(got, oracle.poseidon_merkle(po, oracles, cs, additive, salts))


def check_leaf_and_two_to_one_kats(lib):
    # a 2-leaf tree over one oracle with coset_size c has leaves = leafhash(slice) and root = two_to_one(l, r):
    # LeafTest / two-to-one known answers of test_poseidon.cpp:76-80,103-119 through the Merkle entry point
    p, po = param_pair("test_params")
    zero = oracle.bn_from_ints([0, 0])
    nodes = lib.merkle_tree_poseidon(p, [zero], 1, 1)
    assert oracle.bn_to_ints(nodes[1:2])[0] == KAT["zero_state_squeeze_test_params"]
    assert np.array_equal(nodes[0], oracle.poseidon_two_to_one(po, nodes[1], nodes[2]))


def check_errors(lib):
    import pytest
    p, _ = param_pair("test_params")
    with pytest.raises(ValueError):         # merkle_tree.tcc:27-31
        lib.merkle_tree_poseidon(p, [rand_bn(1, 6)], 2)
    with pytest.raises(AssertionError):     # merkle_tree.tcc:98-108
        lib.merkle_tree_poseidon(p, [rand_bn(1, 9)], 2)
    bad = libiop_amd.PoseidonParams.from_dict(dict(KAT["test_params"], alpha=7))
    with pytest.raises(ValueError):
        lib.poseidon_permute(bad, np.zeros((1, 3, 4), dtype=np.uint64))
