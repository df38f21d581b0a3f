// Prime field F_p, p = 1552511030102430251236801561344621993261920897571225601 (181 bits: libff's edwards_Fr,
// the field of the reference's --field_size 181 runs), on gfx950.
//
// Elements are Montgomery representatives a * 2^192 mod p stored as three little-endian uint64 words — libff
// Fp_model's `mont_repr`, which libiop hashes and samples raw (libiop/bcs/hashing/blake2b.tcc:148-152,197-227) —
// so HBM buffers are byte-compatible with the reference's std::vector<FieldT>.  On the device a stored element is six
// 32-bit words (additions, subtractions); products run on seven 29-bit limbs (see "products: radix 2^29" below).
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

// ---- products: radix 2^29 -------------------------------------------------------------------------------------------
// gfx950's integer multiplier is v_mad_u64_u32 (32 x 32 + 64 -> 64, no carry-in).  With seven 29-bit limbs a whole column of
// a Montgomery product (7 a_i * b_j + 7 m_i * p_j terms, each < 2^60) fits the 64-bit accumulator, so a product is 49 + 42
// multiply-adds and no carry chains (the 6 x 32-bit CIOS form compiled to 77 multiply-adds plus ~320 carry / move ops).
// The Montgomery radix of this form is 2^203, not libff's 2^192.  Every product in this library is DATA x TABLE (twiddle,
// scale factor, fold or combination coefficient) or TABLE x TABLE, never data x data, so: data stays in libff's form
// (x * 2^192), tables are kept as t * 2^203 ("table form", hfp3::table_form()), and
//     fp_mul(data, table) = (x 2^192)(t 2^203) / 2^203 = (x t) 2^192      data again,
//     fp_mul(table, table) = (s 2^203)(t 2^203) / 2^203 = (s t) 2^203     table again.
struct fp7 {
    uint32_t l[7];
};

#define FP7_MASK 0x1fffffffu
__device__ static const uint32_t FP7_P[7] = { 0x00000001u, 0x0f2a993cu, 0x0b84b307u, 0x05c92772u, 0x08e56c4eu, 0x1abf93a5u, 0x00000040u };
#define FP7_INV 0x1fffffffu             /* -p^{-1} mod 2^29 (p = 1 mod 2^29) */

// 192-bit value -> seven 29-bit limbs (the top limb takes bits 174..191)
__device__ __forceinline__ fp7 fp7_unpack(const fp3 &a)
{
    fp7 r;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)a.w[wi] | ((uint64_t)(wi + 1 < 6 ? a.w[wi + 1] : 0u) << 32);
        r.l[i] = (uint32_t)(two >> sh) & FP7_MASK;
    }
    return r;
}

// normalised limbs (value < 2^192) -> six 32-bit words
__device__ __forceinline__ fp3 fp7_pack(const fp7 &y)
{
    fp3 r;
    uint64_t acc = 0;
    int bits = 0, wi = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        acc |= (uint64_t)y.l[i] << bits;
        bits += 29;
        if (bits >= 32 && wi < 6) { r.w[wi++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
    }
    return r;                   // 7 * 29 = 203 bits: six words emitted, the 11 bits above them are zero for values < 2^192
}

// a * b * 2^-203 mod p, result < 2p with normalised limbs.  Column sums stay below 2^64 for operand limbs up to 2^30 (both)
// or 2^31 against a normalised operand; values up to 2^192 on both sides keep the result below 2p.
__device__ __forceinline__ fp7 fp7_mul(const fp7 &a, const fp7 &b)
{
    uint32_t m[7];
    fp7 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * FP7_P[k - i];
        m[k] = ((uint32_t)acc * FP7_INV) & FP7_MASK;
        acc += (uint64_t)m[k] * FP7_P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 7; k < 13; ++k) {
#pragma unroll
        for (int i = k - 6; i < 7; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 6; i < 7; ++i) acc += (uint64_t)m[i] * FP7_P[k - i];
        r.l[k - 7] = (uint32_t)acc & FP7_MASK;
        acc >>= 29;
    }
    r.l[6] = (uint32_t)acc;
    return r;
}

// ---- sums of products with ONE Montgomery reduction ------------------------------------------------------------------------
// A product's 13 column sums are each below 7 * 2^58; up to 8 products can share the 64-bit column accumulators (8 * 7 * 2^58, plus
// the reduction's own 7 * 2^58 per column and a 35-bit carry, stays below 2^64), so sum_i a_i b_i costs 49 multiply-adds per term
// and 42 once, instead of 91 per term.  Operand limbs must be normalised (< 2^29): fp7_unpack of stored values, table constants.
struct fp7w {
    uint64_t c[13];
};

__device__ __forceinline__ void fp7w_zero(fp7w &w)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) w.c[k] = 0;
}

__device__ __forceinline__ void fp7w_mac(fp7w &w, const fp7 &a, const fp7 &b)
{
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) w.c[i + j] += (uint64_t)a.l[i] * b.l[j];
}

// (the accumulated sum) * 2^-203 mod p, below 2p with normalised limbs when at most 8 products of values below 2^192 and p were summed
__device__ __forceinline__ fp7 fp7w_redc(const fp7w &w)
{
    uint32_t m[7];
    fp7 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        acc += w.c[k];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * FP7_P[k - i];
        m[k] = ((uint32_t)acc * FP7_INV) & FP7_MASK;
        acc += (uint64_t)m[k] * FP7_P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 7; k < 13; ++k) {
        acc += w.c[k];
#pragma unroll
        for (int i = k - 6; i < 7; ++i) acc += (uint64_t)m[i] * FP7_P[k - i];
        r.l[k - 7] = (uint32_t)acc & FP7_MASK;
        acc >>= 29;
    }
    r.l[6] = (uint32_t)acc;
    return r;
}

__device__ __forceinline__ void fp_mac(fp7w &w, const fp3 &a, const fp3 &b)
{
    fp7w_mac(w, fp7_unpack(a), fp7_unpack(b));
}

// canonical stored form of the accumulated sum * 2^-203
__device__ __forceinline__ fp3 fp_redc(const fp7w &w)
{
    fp3 r = fp7_pack(fp7w_redc(w));
    fp_cond_sub_p(r.w);
    return r;
}

// p - a for a canonical a != 0, 0 for 0
__device__ __forceinline__ fp3 fp_neg(const fp3 &a)
{
    return fp_sub(fp_zero(), a);
}

// ---- lazily reduced butterflies ---------------------------------------------------------------------------------------
// A radix-2 butterfly (x, y) -> (x + w y, x - w y) on limb vectors with NO carry handling: t = w y is below 2p with 29-bit
// limbs; the difference is formed as x + (8p - t) with 8p written so that each of its lower six limbs is at least 2^29 - 1
// (and the top one above t's), so every limb stays non-negative.  Limbs grow by at most 1.5 * 2^29 per level — below 2^32 after
// the three levels a lane runs on its registers — and fp7_mul takes one operand with any 32-bit limbs against a normalised
// twiddle (7 * 2^32 * 2^29 + 7 * 2^58 < 2^64).  Values grow by at most 8p per level: below 2^189 for any transform length
// of this field (2-adicity 31), so they fit the 192-bit stored form; a transform's last pass makes them canonical.
__device__ static const uint32_t FP7_8P_SPREAD[7] = { 0x20000008u, 0x3954c9dfu, 0x3c25983au, 0x2e493b91u, 0x272b6270u, 0x35fc9d29u, 0x00000205u };
__device__ static const uint32_t FP7_ONE_T[7] = { 0x1f81a675u, 0x0910f06bu, 0x1f077a07u, 0x02c7785du, 0x1a4c3b6au, 0x144d5829u, 0x00000011u };  // 2^203 mod p

__device__ __forceinline__ void fp7_bfly(fp7 &x, fp7 &y, const fp7 &w)
{
    const fp7 t = fp7_mul(w, y);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        y.l[i] = x.l[i] + (FP7_8P_SPREAD[i] - t.l[i]);
        x.l[i] = x.l[i] + t.l[i];
    }
}

// carry propagation: same value, limbs 0..5 back below 2^29
__device__ __forceinline__ fp7 fp7_norm(const fp7 &a)
{
    fp7 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t v = a.l[i] + c;
        r.l[i] = v & FP7_MASK;
        c = v >> 29;
    }
    r.l[6] = a.l[6] + c;
    return r;
}

// any value below 2^192 -> canonical stored form (one product with the table form of 1)
__device__ __forceinline__ fp3 fp7_canonical(const fp7 &v)
{
    fp7 one;
#pragma unroll
    for (int i = 0; i < 7; ++i) one.l[i] = FP7_ONE_T[i];
    fp3 r = fp7_pack(fp7_mul(v, one));
    fp_cond_sub_p(r.w);
    return r;
}

// data x table (or table x table) product on the stored form: canonical in, canonical out
__device__ __forceinline__ fp3 fp_mul(const fp3 &a, const fp3 &b)
{
    fp3 r = fp7_pack(fp7_mul(fp7_unpack(a), fp7_unpack(b)));
    fp_cond_sub_p(r.w);
    return r;
}
