"""The half-wavefront comb product (gf_mul_halves, libiop_amd/csrc/include/iopx/gfx950_comb.h: one multiplier per 32 lanes, EXEC narrowed to one half at
a time inside the asm block and restored) through its test entry iopx_gf192_mul_halves_dev: products against the oracle's field multiplication,
elements outside the branch unchanged, and — on the GPU — the number of lanes a ballot right after the product sees active."""
import numpy as np

import oracle
from helpers import rand_elems

W = 3


def check(lib, torch, count_active=True):
    n = 64 * 24
    a = rand_elems(0x4a1, n, W)
    c2 = rand_elems(0x4a2, 2, W)
    for lane_mask in (0xFFFFFFFF, 1, 2, 0x21, 0x20, 0x3F, 64):
        idx = np.arange(n, dtype=np.uint64)
        taken = (idx & np.uint64(lane_mask & 0xFFFFFFFF)) != 0
        d_a, d_c, d_o, d_act = lib.malloc(n * 24), lib.malloc(48), lib.malloc(n * 24), lib.malloc(4 * (n // 64))
        try:
            lib.h2d(d_a, a)
            lib.h2d(d_c, c2)
            lib.h2d(d_act, np.zeros(n // 64, dtype=np.uint32))
            lib.gf192_mul_halves_dev(d_a, d_c, d_o, d_act, n, lane_mask)
            out = np.empty((n, W), dtype=np.uint64)
            act = np.empty(n // 64, dtype=np.uint32)
            lib.d2h(out, d_o)
            lib.d2h(act, d_act)
        finally:
            for p in (d_a, d_c, d_o, d_act):
                lib.free(p)
        low = (idx % 64) < 32
        want = a.copy()
        sel = taken & low
        want[sel] = oracle.gf_mul(a[sel], np.repeat(c2[0:1], int(sel.sum()), axis=0))
        sel = taken & ~low
        want[sel] = oracle.gf_mul(a[sel], np.repeat(c2[1:2], int(sel.sum()), axis=0))
        assert np.array_equal(out, want), hex(lane_mask)
        if count_active:
            per_wave = taken.reshape(-1, 64).sum(axis=1)
            assert np.array_equal(act, per_wave.astype(np.uint32)), (hex(lane_mask), act, per_wave)
