// Prime field F_p, p = 1552511030102430251236801561344621993261920897571225601 (181 bits: libff's edwards_Fr,
// the field of the reference's --field_size 181 runs), on gfx950.
//
// Elements are Montgomery representatives a * 2^192 mod p stored as three little-endian uint64 words — libff
// Fp_model's `mont_repr`, which libiop hashes and samples raw (libiop/bcs/hashing/blake2b.tcc:148-152,197-227) —
// so HBM buffers are byte-compatible with the reference's std::vector<FieldT>.  On the device an element is six
// 32-bit limbs; products are CIOS Montgomery multiplications on v_mad_u64_u32 (measured 32 T lane-ops/s on
// MI355X, tools/ubench/valu_rates.hip): 72 multiply-adds per field product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct fp3 {
    uint32_t w[6];
};

__device__ static const uint32_t FP3_P[6] = { 0x80000001u, 0x1de55327u, 0xb92e12ccu, 0xc4e2e493u, 0x274a8e56u, 0x0010357fu };
#define FP3_INV32 0x7fffffffu           /* -p^{-1} mod 2^32 */

__device__ __forceinline__ fp3 fp_load(const uint64_t *__restrict__ p, size_t idx)
{
    const uint64_t *q = p + 3 * idx;
    const uint64_t a = q[0], b = q[1], c = q[2];
    fp3 r;
    r.w[0] = (uint32_t)a; r.w[1] = (uint32_t)(a >> 32);
    r.w[2] = (uint32_t)b; r.w[3] = (uint32_t)(b >> 32);
    r.w[4] = (uint32_t)c; r.w[5] = (uint32_t)(c >> 32);
    return r;
}

__device__ __forceinline__ void fp_store(uint64_t *__restrict__ p, size_t idx, const fp3 &v)
{
    uint64_t *q = p + 3 * idx;
    q[0] = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32);
    q[1] = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32);
    q[2] = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
}

__device__ __forceinline__ fp3 fp_zero()
{
    fp3 r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.w[i] = 0;
    return r;
}

// r = a - p if a >= p (a < 2p)
__device__ __forceinline__ void fp_cond_sub_p(uint32_t (&a)[6])
{
    uint32_t d[6];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint64_t t = (uint64_t)a[i] - FP3_P[i] - borrow;
        d[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = d[i];
    }
}

__device__ __forceinline__ fp3 fp_add(const fp3 &a, const fp3 &b)
{
    fp3 r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint64_t t = (uint64_t)a.w[i] + b.w[i] + carry;
        r.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    fp_cond_sub_p(r.w);         // a + b < 2p < 2^192: no carry out
    return r;
}

__device__ __forceinline__ fp3 fp_sub(const fp3 &a, const fp3 &b)
{
    fp3 r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint64_t t = (uint64_t)a.w[i] - b.w[i] - borrow;
        r.w[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    if (borrow) {
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint64_t t = (uint64_t)r.w[i] + FP3_P[i] + carry;
            r.w[i] = (uint32_t)t;
            carry = t >> 32;
        }
    }
    return r;
}

// Montgomery product a * b * 2^-192 mod p (CIOS, 32-bit limbs)
__device__ __forceinline__ fp3 fp_mul(const fp3 &a, const fp3 &b)
{
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint64_t cur = (uint64_t)a.w[j] * b.w[i] + t[j] + carry;
            t[j] = (uint32_t)cur;
            carry = cur >> 32;
        }
        uint64_t cur = (uint64_t)t[6] + carry;
        t[6] = (uint32_t)cur;
        t[7] = (uint32_t)(cur >> 32);
        const uint32_t m = t[0] * FP3_INV32;
        cur = (uint64_t)m * FP3_P[0] + t[0];
        carry = cur >> 32;
#pragma unroll
        for (int j = 1; j < 6; ++j) {
            cur = (uint64_t)m * FP3_P[j] + t[j] + carry;
            t[j - 1] = (uint32_t)cur;
            carry = cur >> 32;
        }
        cur = (uint64_t)t[6] + carry;
        t[5] = (uint32_t)cur;
        t[6] = t[7] + (uint32_t)(cur >> 32);
    }
    fp3 r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.w[i] = t[i];
    fp_cond_sub_p(r.w);         // the CIOS result is < 2p and p < 2^181, so t[6] is always zero
    return r;
}
