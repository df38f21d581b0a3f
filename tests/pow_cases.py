"""Proof-of-work parity cases shared by the CPU-emulation suite and the GPU suite (pow.tcc:67-162)."""
import numpy as np
import pytest

import oracle
import poseidon_cases as pc


def check_blake2b(lib, bitlens, seeds):
    for seed in seeds:
        ch = bytes(np.random.default_rng(seed).integers(0, 256, size=32, dtype=np.uint8))
        for bitlen in bitlens:
            want, calls = oracle.pow_solve_blake2b(ch, bitlen)
            got = lib.solve_pow(ch, bitlen)
            assert got == want, (seed, bitlen, calls)
            assert oracle.pow_verify_blake2b(ch, got, bitlen)       # test_pow.cpp:31-32


def check_blake2b_reference_test_inputs(lib):
    # test_pow.cpp:13-33: log_work 20, cost 1, challenge "abcdefghijklmnopqrstuvwxyzabcdef"
    ch = b"abcdefghijklmnopqrstuvwxyzabcdef"
    bitlen = oracle.pow_bitlen(20, 1)
    assert bitlen == 20
    got = lib.solve_pow(ch, bitlen)
    assert oracle.pow_verify_blake2b(ch, got, bitlen)
    assert got == oracle.pow_solve_blake2b(ch, bitlen)[0]


def check_challenge_itself_passes(lib):
    # bitlen 0: the mask is empty, candidate 0 (the challenge, untouched) is the answer (pow.tcc:92-96)
    ch = bytes(range(32))
    assert lib.solve_pow(ch, 0) == ch


def check_poseidon(lib, name, bitlens, seeds):
    p, po = pc.param_pair(name)
    for seed in seeds:
        ch = pc.rand_bn(1000 + seed, 4)[3]
        for bitlen in bitlens:
            want, calls = oracle.pow_solve_poseidon(po, ch, bitlen)
            got = lib.solve_pow(ch, bitlen, poseidon_params=p)
            assert np.array_equal(got, want), (name, seed, bitlen, calls)
            assert oracle.pow_verify_poseidon(po, ch, got, bitlen)  # test_pow.cpp:53-54


def check_errors(lib):
    with pytest.raises(ValueError):
        lib.solve_pow(bytes(32), 31)


def test_bitlen_rule():
    # pow_parameters::pow_bitlen (pow.tcc:21-32) and the defaults of common_bcs_parameters.tcc:23-25
    assert oracle.pow_bitlen(20, 1) == 20
    assert oracle.pow_bitlen(20, 200) == 13          # test_pow.cpp:41-44: floor(log2 200) = 7
    assert oracle.pow_bitlen(20 + 3 + 7, 128) == 23  # dim_h = 20, algebraic hash
    assert oracle.pow_bitlen(20 + 3 + 0, 1) == 23    # dim_h = 20, BLAKE2b


def check_search_in_two_halves(lib):
    """iopx_pow_search_blake2b_begin / _end: the same answer as the one-call search; misuse (a second begin, an end without a begin) is refused."""
    import pytest
    ch = bytes(range(32))
    whole = lib.pow_search(ch, 9, 0, 1 << 14)
    lib.pow_search_begin(ch, 9, 0, 1 << 14)
    with pytest.raises(AssertionError):
        lib.pow_search_begin(ch, 9, 0, 16)          # IOPX_ERR_LOGIC: one pending search at a time
    assert lib.pow_search_end() == whole and whole is not None
    with pytest.raises(AssertionError):
        lib.pow_search_end()
    lib.pow_search_begin(ch, 30, 5, 0)               # an empty range: nothing found
    assert lib.pow_search_end() is None
