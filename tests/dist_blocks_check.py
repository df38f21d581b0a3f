"""Shared checks of the sharded-transform building blocks (taylor / pow_table / combine) against numpy + oracle
primitives; run on the CPU emulation (tests/test_kernel_logic_emu.py) and on the GPU (tests/test_gpu_parity.py)."""
import numpy as np

import oracle
from helpers import rand_elems
from libiop_amd import host

W = 3


def _dev(lib, arr):
    p = lib.malloc(arr.nbytes)
    lib.h2d(p, arr)
    return p


def _back(lib, p, shape):
    out = np.empty(shape, dtype=np.uint64)
    lib.d2h(out, p)
    lib.free(p)
    return out


def check_pow_table(lib, count=300):
    base, init = rand_elems(1, 1, W)[0], rand_elems(2, 1, W)[0]
    p = lib.malloc(count * 24)
    lib.pow_table_dev(p, count, base, init)
    lib.synchronize()
    got = _back(lib, p, (count, W))
    cur, b = host.gf_from_words(init), host.gf_from_words(base)
    for l in range(count):
        assert host.gf_from_words(got[l]) == cur, l
        cur = host.gf_mul(cur, b)


def check_taylor(lib, log_n):
    n = 1 << log_n
    S, tw = rand_elems(3 + log_n, n, W), rand_elems(4 + log_n, n, W)
    for use_twist in (True, False):
        ref = oracle.gf_mul(S, tw) if use_twist else S.copy()
        stride = n // 4
        while stride >= 1:                      # fft.tcc:73-83 with j = 0
            for ofs in range(0, n, stride * 4):
                ref[ofs + 2 * stride:ofs + 3 * stride] ^= ref[ofs + 3 * stride:ofs + 4 * stride]
                ref[ofs + stride:ofs + 2 * stride] ^= ref[ofs + 2 * stride:ofs + 3 * stride]
            stride //= 2
        dS, dT = _dev(lib, S), _dev(lib, tw)
        lib.taylor_dev(dS, log_n, dT if use_twist else 0)
        lib.synchronize()
        lib.free(dT)
        assert np.array_equal(_back(lib, dS, (n, W)), ref), (log_n, use_twist)


def check_combine(lib, count=1000, nb=13, index_base=4096):
    a, b = rand_elems(5, count, W), rand_elems(6, count, W)
    B, sh = rand_elems(7, nb, W), rand_elems(8, 1, W)[0]
    tw = np.repeat(sh[None, :], count, axis=0)
    for i in range(count):
        idx = index_base + i
        for k in range(nb):
            if (idx >> k) & 1:
                tw[i] ^= B[k]
    lower = a ^ oracle.gf_mul(b, tw)
    for upper in (0, 1):
        da, db = _dev(lib, a), _dev(lib, b)
        do = lib.malloc(count * 24)
        lib.combine_dev(da, db, do, count, index_base, B, sh, upper)
        lib.synchronize()
        lib.free(da)
        lib.free(db)
        assert np.array_equal(_back(lib, do, (count, W)), lower ^ b if upper else lower), upper
