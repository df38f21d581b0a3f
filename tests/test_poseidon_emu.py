"""Poseidon kernels' logic on the CPU: the product .hip source compiled by tests/emu (no GPU), called through the C ABI and
compared with the oracle and the reference's known answers.  The same cases run on the MI355X in test_gpu_parity.py."""
import pytest

import poseidon_cases as pc
from emu_lib import emu


def test_to_montgomery():
    pc.check_to_montgomery(emu())


def test_permutation_kats():
    pc.check_permutation_kats(emu())


@pytest.mark.parametrize("name", pc.SET_NAMES)
def test_permutation(name):
    pc.check_permutation(emu(), name, count=5)


@pytest.mark.parametrize("name,r,cs,L,additive,zk", [
    ("test_params", 1, 1, 2, False, False), ("test_params", 1, 2, 8, False, True), ("test_params", 3, 2, 4, True, False),
    ("starkware_alpha5_t3", 2, 4, 4, False, False), ("high_alpha17_t3", 1, 2, 8, False, True), ("high_alpha17_t4", 2, 3, 4, False, True),
    ("high_alpha17_t4", 1, 6, 2, True, False), ("test_params", 1, 1, 1024, False, False),
])
def test_merkle(name, r, cs, L, additive, zk):
    pc.check_merkle(emu(), name, r, cs, L, additive, zk)


def test_leaf_and_two_to_one_kats():
    pc.check_leaf_and_two_to_one_kats(emu())


def test_errors():
    pc.check_errors(emu())
