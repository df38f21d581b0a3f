"""Transcript extraction on the CPU (product .hip sources compiled by tests/emu) against the oracle."""
import pytest

import transcript_cases as tc
from emu_lib import emu
from transcript_cases import test_hash_count_expectation  # noqa: F401


def test_membership_proofs():
    tc.check_membership_proofs(emu(), 64, 1, [1, 2, 5, 17, 64, 200])
    tc.check_membership_proofs(emu(), 2, 2, [1, 2, 3])


def test_all_subsets_of_small_tree():
    tc.check_all_subsets_of_small_tree(emu())


def test_empty_and_errors():
    tc.check_empty_and_errors(emu())


def test_deferred_downloads():
    tc.check_deferred_downloads(emu())


def test_large_odd_deferred_downloads():
    tc.check_large_odd_deferred_downloads(emu())


def test_query_responses():
    tc.check_query_responses(emu(), 256, 3, 5)


@pytest.mark.parametrize("m,d,batch", [(6, 3, 1), (9, 5, 3), (8, 8, 2), (10, 1, 2), (12, 7, 2)])
def test_reextend_equals_ifft_then_fft(m, d, batch):
    import torch
    tc.check_reextend(emu(), torch, torch.device("cpu"), m, d, batch, 40 + m)
