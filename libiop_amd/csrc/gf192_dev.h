// GF(2^192) = GF(2)[x]/(x^192 + x^7 + x^2 + x + 1) on gfx950 (CDNA4).
//
// gfx950 has no carry-less multiply, so the field product is built from 32-bit VALU logic ops.
// Element layout in HBM is the reference's in-memory layout (libff gf192: three little-endian
// uint64 words, polynomial basis) — 24 bytes, 8-byte aligned — so that buffers can be hashed and
// exchanged with the host byte-for-byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct gf192 {
    uint32_t w[6];
};

__device__ __forceinline__ gf192 gf_zero()
{
    gf192 r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.w[i] = 0;
    return r;
}

__device__ __forceinline__ gf192 gf_load(const uint64_t *__restrict__ p, size_t idx)
{
    const uint64_t *q = p + 3 * idx;
    const uint64_t a = q[0], b = q[1], c = q[2];
    gf192 r;
    r.w[0] = (uint32_t)a; r.w[1] = (uint32_t)(a >> 32);
    r.w[2] = (uint32_t)b; r.w[3] = (uint32_t)(b >> 32);
    r.w[4] = (uint32_t)c; r.w[5] = (uint32_t)(c >> 32);
    return r;
}

__device__ __forceinline__ void gf_store(uint64_t *__restrict__ p, size_t idx, const gf192 &v)
{
    uint64_t *q = p + 3 * idx;
    q[0] = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32);
    q[1] = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32);
    q[2] = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
}

__device__ __forceinline__ gf192 gf_add(const gf192 &a, const gf192 &b)
{
    gf192 r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] ^ b.w[i];
    return r;
}

__device__ __forceinline__ void gf_add_to(gf192 &a, const gf192 &b)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) a.w[i] ^= b.w[i];
}

__device__ __forceinline__ bool gf_is_zero(const gf192 &a)
{
    return (a.w[0] | a.w[1] | a.w[2] | a.w[3] | a.w[4] | a.w[5]) == 0;
}

// Fold a 12-word (383-bit) carry-less product modulo x^192 + x^7 + x^2 + x + 1.
// x^192 = x^7 + x^2 + x + 1, so word i (i >= 6) folds onto words i-6 and i-5.
__device__ __forceinline__ gf192 gf_reduce(uint32_t (&c)[12])
{
#pragma unroll
    for (int i = 11; i >= 6; --i) {
        const uint32_t t = c[i];
        c[i - 6] ^= t ^ (t << 1) ^ (t << 2) ^ (t << 7);
        c[i - 5] ^= (t >> 31) ^ (t >> 30) ^ (t >> 25);
    }
    gf192 r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.w[i] = c[i];
    return r;
}

// c ^ (m & b) in one VALU op: gfx950's three-input bitwise op (truth table over a=0xF0, b=0xCC, c=0xAA)
__device__ __forceinline__ uint32_t xor_and(uint32_t c, uint32_t m, uint32_t b)
{
    return __builtin_amdgcn_bitop3_b32(c, m, b, 0x78);
}

// General product, both operands per-lane.  Horner over the bit position inside a word: the 12-word
// accumulator is shifted left by one per step and, for each of the six words of a, the bit at that
// position (sign-extended by v_bfe_i32) conditionally adds b at that word offset with one v_bitop3
// per word: 32 x (12 + 6 + 36) = ~1.7k VALU ops.
__device__ __forceinline__ gf192 gf_mul(const gf192 &a, const gf192 &b)
{
    uint32_t c[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) c[i] = 0;
#pragma unroll 4
    for (int t = 31; t >= 0; --t) {
#pragma unroll
        for (int i = 11; i > 0; --i) c[i] = (c[i] << 1) | (c[i - 1] >> 31);
        c[0] <<= 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)a.w[k], t, 1);
#pragma unroll
            for (int w = 0; w < 6; ++w) c[k + w] = xor_and(c[k + w], m, b.w[w]);
        }
    }
    return gf_reduce(c);
}
