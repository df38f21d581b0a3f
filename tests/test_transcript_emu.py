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
    tc.check_query_responses(emu(), 4096, 2, 6, num_positions=400)          # more than 256 positions: device arrays
    tc.check_query_responses(emu(), 128, 17, 7)                             # more than 16 oracles: device arrays


def test_wide_tree():
    tc.check_wide_tree(emu())


@pytest.mark.parametrize("m,d,batch", [(6, 3, 1), (9, 5, 3), (8, 8, 2), (10, 1, 2), (12, 7, 2)])
def test_reextend_equals_ifft_then_fft(m, d, batch):
    import torch
    tc.check_reextend(emu(), torch, torch.device("cpu"), m, d, batch, 40 + m)


# ---- the small device helpers of the multi-GPU layer (include/libiop_amd.h "multi-GPU"; libiop_amd/csrc/comm.hip) ----
def test_interleave_and_gather_rows():
    import ctypes
    import numpy as np
    lib = emu()
    rng = np.random.default_rng(3)
    parts, count = 4, 37
    src = rng.integers(0, 2**63, size=(parts, count, 3), dtype=np.uint64)          # rank-major residue classes
    d_src, d_dst = lib.malloc(src.nbytes), lib.malloc(src.nbytes)
    try:
        lib.h2d(d_src, src)
        lib.c.iopx_interleave_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        lib._check(lib.c.iopx_interleave_dev(d_src, parts, count, 24, d_dst))
        out = np.empty((count * parts, 3), dtype=np.uint64)
        lib.d2h(out, d_dst)
        assert np.array_equal(out, src.transpose(1, 0, 2).reshape(-1, 3))            # element i * parts + r = class r's entry i
        with pytest.raises(ValueError):
            lib._check(lib.c.iopx_interleave_dev(d_src, parts, count, 24, d_src))    # in place is refused
        with pytest.raises(ValueError):
            lib._check(lib.c.iopx_interleave_dev(d_src, parts, count, 12, d_dst))    # element size must be a multiple of 8
    finally:
        lib.free(d_src); lib.free(d_dst)
    # gather_rows: out[dst_row[i]][k] = srcs[k][src_index[i]], other rows untouched (zero)
    n, num, rows = 50, 3, 9
    srcs = [rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) for _ in range(num)]     # 32-byte elements (digests)
    d_srcs = [lib.malloc(a.nbytes) for a in srcs]
    d_out = lib.malloc(rows * num * 32)
    try:
        for d, a in zip(d_srcs, srcs):
            lib.h2d(d, a)
        lib.h2d(d_out, np.zeros((rows, num, 4), dtype=np.uint64))
        src_index = np.array([49, 0, 7, 7], dtype=np.uint64)
        dst_row = np.array([8, 2, 0, 5], dtype=np.uint64)
        ptrs = (ctypes.c_void_p * num)(*d_srcs)
        lib.c.iopx_gather_rows_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        lib._check(lib.c.iopx_gather_rows_dev(ptrs, num, 32, src_index.ctypes.data, dst_row.ctypes.data, len(src_index), d_out))
        out = np.empty((rows, num, 4), dtype=np.uint64)
        lib.d2h(out, d_out)
        want = np.zeros_like(out)
        for i, r in zip(src_index, dst_row):
            for k in range(num):
                want[int(r), k] = srcs[k][int(i)]
        assert np.array_equal(out, want)
    finally:
        for d in d_srcs:
            lib.free(d)
        lib.free(d_out)


def test_communicator_argument_checks():
    import ctypes
    lib = emu()
    h = ctypes.c_void_p()
    lib.c.iopx_comm_create_callbacks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    cbs = (ctypes.c_void_p * 6)()                                                        # all null: allowed for one rank only
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_comm_create_callbacks(0, 3, ctypes.addressof(cbs), ctypes.byref(h)))      # world must be a power of two
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_comm_create_callbacks(2, 2, ctypes.addressof(cbs), ctypes.byref(h)))      # rank outside the world
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_comm_create_callbacks(0, 2, ctypes.addressof(cbs), ctypes.byref(h)))      # collectives missing
    lib._check(lib.c.iopx_comm_create_callbacks(0, 1, ctypes.addressof(cbs), ctypes.byref(h)))
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    lib.c.iopx_comm_rank.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib._check(lib.c.iopx_comm_rank(h, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (0, 1)
    lib.c.iopx_comm_all_gather_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_comm_all_gather_dev(None, None, None, 8))                  # null communicator
    lib.comm_destroy(h)
