"""Proof-of-work kernels' logic on the CPU (product .hip sources compiled by tests/emu), against the oracle."""
import pow_cases as pw
from emu_lib import emu
from pow_cases import test_bitlen_rule  # noqa: F401  (oracle-only case, runs in the CPU suite)


def test_blake2b():
    pw.check_blake2b(emu(), [0, 1, 4, 9, 12], [1, 2, 3])


def test_blake2b_crosses_batches():
    pw.check_blake2b(emu(), [17], [7])       # > 2^16 candidates: the answer lies beyond the first launch


def test_challenge_itself_passes():
    pw.check_challenge_itself_passes(emu())


def test_poseidon():
    pw.check_poseidon(emu(), "test_params", [0, 3, 8], [1, 2])
    pw.check_poseidon(emu(), "high_alpha17_t3", [6], [3])
    pw.check_poseidon(emu(), "high_alpha17_t4", [5], [4])


def test_errors():
    pw.check_errors(emu())


def test_search_in_two_halves():
    pw.check_search_in_two_halves(emu())
