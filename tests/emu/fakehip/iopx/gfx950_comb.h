// TEST INFRASTRUCTURE ONLY: C++ model of libiop_amd/csrc/include/iopx/gfx950_comb.h for the CPU emulation
// (same algorithm: 4-bit-window comb over the nibbles of c with a table of a*u).
#pragma once
#include <stdint.h>
static inline void comb_clmul_192_uniform(uint32_t (&r)[12], const uint32_t (&a)[6], const uint32_t (&c)[6])
{
    uint32_t T[16][7];
    for (int i = 0; i < 7; ++i) T[0][i] = 0;
    for (int i = 0; i < 6; ++i) T[1][i] = a[i];
    T[1][6] = 0;
    for (int u = 2; u < 16; u += 2) {
        for (int i = 6; i > 0; --i) T[u][i] = (T[u >> 1][i] << 1) | (T[u >> 1][i - 1] >> 31);
        T[u][0] = T[u >> 1][0] << 1;
        for (int i = 0; i < 7; ++i) T[u + 1][i] = T[u][i] ^ T[1][i];
    }
    for (int i = 0; i < 12; ++i) r[i] = 0;
    for (int o = 7; o >= 0; --o) {
        if (o != 7) {
            for (int i = 11; i > 0; --i) r[i] = (r[i] << 4) | (r[i - 1] >> 28);
            r[0] <<= 4;
        }
        for (int k = 0; k < 6; ++k) {
            const uint32_t nib = (c[k] >> (4 * o)) & 15u;
            for (int i = 0; i < 7 && k + i < 12; ++i) r[k + i] ^= T[nib][i];
        }
    }
}

// the halves form: the emulation runs one lane at a time, so the caller's lane picks the multiplier its half uses (iopx_emu_lane, set by gf_mul_halves)
extern thread_local int iopx_emu_lane;
static inline void iopx_set_emu_lane(int lane) { iopx_emu_lane = lane; }
static inline void comb_clmul_192_halves(uint32_t (&r)[12], const uint32_t (&a)[6], const uint32_t (&c)[6], const uint32_t (&d)[6])
{
    comb_clmul_192_uniform(r, a, (iopx_emu_lane & 32) ? d : c);
}

static inline uint64_t uniform_load64(const uint64_t *p) { return *p; }
