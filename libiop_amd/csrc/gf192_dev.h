// GF(2^192) = GF(2)[x]/(x^192 + x^7 + x^2 + x + 1) on gfx950 (CDNA4).
//
// gfx950 has no carry-less multiply, so the field product is built from 32-bit VALU logic ops.
// Element layout in HBM is the reference's in-memory layout (libff gf192: three little-endian
// uint64 words, polynomial basis) — 24 bytes, 8-byte aligned — so that buffers can be hashed and
// exchanged with the host byte-for-byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <iopx/gfx950_comb.h>

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

// Fold a 12-word (383-bit) carry-less product modulo x^192 + x^7 + x^2 + x + 1: with H = c[6..11], the result is
// L + H (1 + x + x^2 + x^7) and a second, 7-bit fold of what that pushes past bit 191.
//
// Written as multi-word shifts of H — word j of H << s is one v_alignbit of (H[j], H[j-1]) — so a word of the result costs three
// shift-class ops and two three-input XORs: 24 slow + 16 fast ops in all.  The word-at-a-time form (for each high word t:
// c[i-6] ^= t ^ t<<1 ^ t<<2 ^ t<<7, c[i-5] ^= t>>31 ^ t>>30 ^ t>>25) takes 36 + 24; shifts issue at 4.2 cycles against 2.5
// (profiles/r03_valu_rates.txt), and the reduction follows every product of every kernel.
__device__ __forceinline__ uint32_t gf_funnel(uint32_t hi, uint32_t lo, int s)          // bits [32 - s, 64 - s) of hi:lo, 0 < s < 32
{
    return __builtin_amdgcn_alignbit(hi, lo, 32 - s);
}

__device__ __forceinline__ gf192 gf_reduce(uint32_t (&c)[12])
{
    gf192 r;
    uint32_t prev = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const uint32_t h = c[6 + j];
        const uint32_t s1 = gf_funnel(h, prev, 1), s2 = gf_funnel(h, prev, 2), s7 = gf_funnel(h, prev, 7);
        r.w[j] = __builtin_amdgcn_bitop3_b32(c[j], h, s1, 0x96);
        r.w[j] = __builtin_amdgcn_bitop3_b32(r.w[j], s2, s7, 0x96);
        prev = h;
    }
    // bits 192..198: the top 1, 2 and 7 bits of H's last word
    const uint32_t t = (prev >> 31) ^ (prev >> 30) ^ (prev >> 25);
    r.w[0] ^= t ^ (t << 1) ^ (t << 2) ^ (t << 7);
    return r;
}

// c ^ (m & b) in one VALU op: gfx950's three-input bitwise op (truth table over a=0xF0, b=0xCC, c=0xAA)
__device__ __forceinline__ uint32_t xor_and(uint32_t c, uint32_t m, uint32_t b)
{
    return __builtin_amdgcn_bitop3_b32(c, m, b, 0x78);
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// bitwise select: (a & m) | (b & ~m)
__device__ __forceinline__ uint32_t bsel(uint32_t a, uint32_t b, uint32_t m)
{
    return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4);
}

// ---------------------------------------------------------------------------------------------------
// General product, both operands per-lane: integer multiplication "with holes" + Karatsuba.
//
// Measured on MI355X (tools/ubench/valu_rates.hip): v_xor/v_and/v_bitop3 57 T lane-ops/s, shifts /
// v_alignbit / v_bfe 36 T, v_mad_u64_u32 32 T — a 32x32->64 integer multiply costs about as much as two
// logic ops, so the carry-less 32x32 product is built from 16 integer multiplies of operands that keep
// only every 4th bit (at most 8 terms meet in a 4-bit slot, so no carry crosses a slot; BearSSL's
// ctmul technique) instead of 32 shift/mask steps.  192x192 = 18 such products (Karatsuba 2-way over
// 96-bit halves, 3-way inside each half).  ~1.0k VALU ops against ~1.7k for the bit-serial form.
// ---------------------------------------------------------------------------------------------------
struct holes4 {
    uint32_t s[4];          // s[i] = x & (0x11111111 << i)
};

__device__ __forceinline__ holes4 holes_split(uint32_t x)
{
    holes4 r;
    r.s[0] = x & 0x11111111u; r.s[1] = x & 0x22222222u; r.s[2] = x & 0x44444444u; r.s[3] = x & 0x88888888u;
    return r;
}

__device__ __forceinline__ holes4 holes_add(const holes4 &a, const holes4 &b)
{
    holes4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.s[i] = a.s[i] ^ b.s[i];
    return r;
}

// carry-less x * y (32x32 -> 64) from the hole-split operands; lo/hi receive the two result words
__device__ __forceinline__ void clmul32_holes(const holes4 &x, const holes4 &y, uint32_t &lo, uint32_t &hi)
{
    uint64_t p[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) p[i][j] = (uint64_t)x.s[i] * (uint64_t)y.s[j];
    uint32_t zl[4], zh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // slot class k collects the products with i + j = k (mod 4)
        const uint64_t a = p[0][k], b = p[1][(k + 3) & 3], c = p[2][(k + 2) & 3], d = p[3][(k + 1) & 3];
        zl[k] = xor3((uint32_t)a, (uint32_t)b, (uint32_t)c) ^ (uint32_t)d;
        zh[k] = xor3((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) ^ (uint32_t)(d >> 32);
    }
    // take class k at bit positions = k (mod 4)
    lo = bsel(bsel(zl[0], zl[1], 0x55555555u), bsel(zl[2], zl[3], 0x55555555u), 0x33333333u);
    hi = bsel(bsel(zh[0], zh[1], 0x55555555u), bsel(zh[2], zh[3], 0x55555555u), 0x33333333u);
}

// 3 x 3 words -> 6 words (Karatsuba with 6 word products); a, b are hole-split words
__device__ __forceinline__ void clmul96(const holes4 (&a)[3], const holes4 (&b)[3], uint32_t (&c)[6])
{
    uint32_t d0l, d0h, d1l, d1h, d2l, d2h, e01l, e01h, e02l, e02h, e12l, e12h;
    clmul32_holes(a[0], b[0], d0l, d0h);
    clmul32_holes(a[1], b[1], d1l, d1h);
    clmul32_holes(a[2], b[2], d2l, d2h);
    clmul32_holes(holes_add(a[0], a[1]), holes_add(b[0], b[1]), e01l, e01h);
    clmul32_holes(holes_add(a[0], a[2]), holes_add(b[0], b[2]), e02l, e02h);
    clmul32_holes(holes_add(a[1], a[2]), holes_add(b[1], b[2]), e12l, e12h);
    // c = d0 + X (e01 + d0 + d1) + X^2 (e02 + d0 + d1 + d2) + X^3 (e12 + d1 + d2) + X^4 d2,  X = x^32
    const uint32_t m1l = xor3(e01l, d0l, d1l), m1h = xor3(e01h, d0h, d1h);
    const uint32_t m2l = xor3(e02l, d0l, d2l) ^ d1l, m2h = xor3(e02h, d0h, d2h) ^ d1h;
    const uint32_t m3l = xor3(e12l, d1l, d2l), m3h = xor3(e12h, d1h, d2h);
    c[0] = d0l;
    c[1] = d0h ^ m1l;
    c[2] = xor3(m1h, m2l, 0u);
    c[3] = xor3(m2h, m3l, 0u);
    c[4] = xor3(m3h, d2l, 0u);
    c[5] = d2h;
}

__device__ __forceinline__ gf192 gf_mul(const gf192 &a, const gf192 &b)
{
    holes4 al[3], ah[3], bl[3], bh[3], am[3], bm[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        al[i] = holes_split(a.w[i]); ah[i] = holes_split(a.w[3 + i]);
        bl[i] = holes_split(b.w[i]); bh[i] = holes_split(b.w[3 + i]);
        am[i] = holes_add(al[i], ah[i]); bm[i] = holes_add(bl[i], bh[i]);
    }
    uint32_t p0[6], p1[6], p2[6];
    clmul96(al, bl, p0);
    clmul96(ah, bh, p2);
    clmul96(am, bm, p1);
    uint32_t c[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) { c[i] = p0[i]; c[6 + i] = p2[i]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) c[3 + i] ^= xor3(p1[i], p0[i], p2[i]);
    return gf_reduce(c);
}

// The same product with a small register footprint (round 5).  gf_mul splits all twelve input words into their hole forms up front and
// forms the Karatsuba sums on the split words: 72 VGPRs before the first multiply, 114 - 118 for a kernel built around it, i.e. four
// wavefronts per SIMD.  Here the Karatsuba sums are formed on the WORDS and every word product splits its own two operands — each split
// form is used by exactly one product, and (x0 ^ x1) & m is one three-input op — so the instruction count stays (937 against 931 VALU ops)
// while the product needs 54 VGPRs: the edge kernels fit 80 registers and run six wavefronts per SIMD.
__device__ __forceinline__ void clmul32_words(uint32_t x, uint32_t y, uint32_t &lo, uint32_t &hi)
{
    const uint32_t xs[4] = { x & 0x11111111u, x & 0x22222222u, x & 0x44444444u, x & 0x88888888u };
    const uint32_t ys[4] = { y & 0x11111111u, y & 0x22222222u, y & 0x44444444u, y & 0x88888888u };
    uint32_t zl[4], zh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t a = (uint64_t)xs[0] * ys[k], b = (uint64_t)xs[1] * ys[(k + 3) & 3], c = (uint64_t)xs[2] * ys[(k + 2) & 3], d = (uint64_t)xs[3] * ys[(k + 1) & 3];
        zl[k] = xor3((uint32_t)a, (uint32_t)b, (uint32_t)c) ^ (uint32_t)d;
        zh[k] = xor3((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) ^ (uint32_t)(d >> 32);
    }
    lo = bsel(bsel(zl[0], zl[1], 0x55555555u), bsel(zl[2], zl[3], 0x55555555u), 0x33333333u);
    hi = bsel(bsel(zh[0], zh[1], 0x55555555u), bsel(zh[2], zh[3], 0x55555555u), 0x33333333u);
}

// (a0 + a1 X + a2 X^2)(b0 + b1 X + b2 X^2), X = x^32: Karatsuba with six word products (the formula of clmul96)
__device__ __forceinline__ void clmul96_words(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t (&c)[6])
{
    uint32_t d0l, d0h, d1l, d1h, d2l, d2h, e01l, e01h, e02l, e02h, e12l, e12h;
    clmul32_words(a0, b0, d0l, d0h);
    clmul32_words(a1, b1, d1l, d1h);
    clmul32_words(a2, b2, d2l, d2h);
    clmul32_words(a0 ^ a1, b0 ^ b1, e01l, e01h);
    clmul32_words(a0 ^ a2, b0 ^ b2, e02l, e02h);
    clmul32_words(a1 ^ a2, b1 ^ b2, e12l, e12h);
    const uint32_t m1l = xor3(e01l, d0l, d1l), m1h = xor3(e01h, d0h, d1h);
    const uint32_t m2l = xor3(e02l, d0l, d2l) ^ d1l, m2h = xor3(e02h, d0h, d2h) ^ d1h;
    const uint32_t m3l = xor3(e12l, d1l, d2l), m3h = xor3(e12h, d1h, d2h);
    c[0] = d0l; c[1] = d0h ^ m1l; c[2] = m1h ^ m2l; c[3] = m2h ^ m3l; c[4] = m3h ^ d2l; c[5] = d2h;
}

__device__ __forceinline__ gf192 gf_mul_lean(const gf192 &a, const gf192 &b)
{
    uint32_t p0[6], p1[6], p2[6];
    clmul96_words(a.w[0], a.w[1], a.w[2], b.w[0], b.w[1], b.w[2], p0);
    clmul96_words(a.w[3], a.w[4], a.w[5], b.w[3], b.w[4], b.w[5], p2);
    clmul96_words(a.w[0] ^ a.w[3], a.w[1] ^ a.w[4], a.w[2] ^ a.w[5], b.w[0] ^ b.w[3], b.w[1] ^ b.w[4], b.w[2] ^ b.w[5], p1);
    uint32_t c[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) { c[i] = p0[i]; c[6 + i] = p2[i]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) c[3 + i] ^= xor3(p1[i], p0[i], p2[i]);
    return gf_reduce(c);
}

// ---------------------------------------------------------------------------------------------------
// Product by  y / x^k  for a one-word y and 0 <= k < 32.
//
// The last butterfly level of the additive FFT multiplies by the points of the domain divided by its last basis vector
// (fft.tcc:55-96 normalises by betas.back()).  Over the standard basis 1, x, ..., x^(d-1) — libiop's default affine subspaces — a
// point is a polynomial of degree < 32 and the divisor is x^(d-1): the product is six one-word carry-less products instead of
// eighteen, and the division by x^k is exact after adding the multiple q P of the modulus that clears the low k bits
// (q = c / (1 + x + x^2 + x^7) mod x^k: P's low part inverted as a power series, five shift-and-add steps).
// About 0.95k issue cycles per wave against 3.25k for gf_mul.
// ---------------------------------------------------------------------------------------------------
// v / x^k in the field, 0 <= k < 32
__device__ __forceinline__ gf192 gf_div_xk(const gf192 &v, int k)
{
    // q = v / (1 + u) mod x^k, u = x + x^2 + x^7:  1 / (1 + u) = (1 + u)(1 + u^2)(1 + u^4)(1 + u^8)(1 + u^16) mod x^32
    uint32_t q = v.w[0];
    q ^= (q << 1) ^ (q << 2) ^ (q << 7);
    q ^= (q << 2) ^ (q << 4) ^ (q << 14);
    q ^= (q << 4) ^ (q << 8) ^ (q << 28);
    q ^= (q << 8) ^ (q << 16);
    q ^= q << 16;
    q &= (1u << k) - 1u;
    // v + q P has k zero low bits; its bits 192.. are q
    const uint32_t c0 = v.w[0] ^ q ^ (q << 1) ^ (q << 2) ^ (q << 7);
    const uint32_t c1 = v.w[1] ^ (q >> 31) ^ (q >> 30) ^ (q >> 25);
    gf192 r;
    r.w[0] = __builtin_amdgcn_alignbit(c1, c0, (uint32_t)k);
    r.w[1] = __builtin_amdgcn_alignbit(v.w[2], c1, (uint32_t)k);
#pragma unroll
    for (int j = 2; j < 5; ++j) r.w[j] = __builtin_amdgcn_alignbit(v.w[j + 1], v.w[j], (uint32_t)k);
    r.w[5] = __builtin_amdgcn_alignbit(q, v.w[5], (uint32_t)k);
    return r;
}

// v / (1 + x) in the field.  P(1) = 1, so v + e P with e = the parity of v is divisible by 1 + x as a polynomial, and the quotient's bit i
// is the XOR of bits 0..i of the dividend: a prefix XOR over the 192 bits, plus — the prefix being linear — the prefix of P's low part
// 1 + x + x^2 + x^7 (= 0x7D, zero from bit 7 on) when e = 1.
__device__ __forceinline__ gf192 gf_div_1px(const gf192 &v)
{
    gf192 r;
    uint32_t carry = 0;                     // all ones when the bits below this word have odd parity
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        uint32_t w = v.w[j];
        w ^= w << 1; w ^= w << 2; w ^= w << 4; w ^= w << 8; w ^= w << 16;
        w ^= carry;
        carry = (uint32_t)((int32_t)w >> 31);
        r.w[j] = w;
    }
    r.w[0] ^= carry & 0x7Du;
    return r;
}

__device__ __forceinline__ gf192 gf_mul_small_over_xk(const gf192 &a, uint32_t y, int k)
{
    const holes4 ys = holes_split(y);
    uint32_t lo[6], hi[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) clmul32_holes(holes_split(a.w[i]), ys, lo[i], hi[i]);
    gf192 v;
    v.w[0] = lo[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) v.w[i] = lo[i] ^ hi[i - 1];
    // bits 192..223 fold back through x^192 = 1 + x + x^2 + x^7
    const uint32_t h = hi[5];
    v.w[0] ^= h ^ (h << 1) ^ (h << 2) ^ (h << 7);
    v.w[1] ^= (h >> 31) ^ (h >> 30) ^ (h >> 25);
    return gf_div_xk(v, k);
}

// Product by  (y1 : y0) / (x^(k1 + k2) (1 + x))  for a two-word numerator and k1, k2 < 32: the second-to-last level over the standard
// basis, whose recursed vectors are (n^2 + n x^k) / x^(2k) for the one-word n of the level above, normalised by the last of them,
// x^(2k - 2) (1 + x) / x^(2k).  Nine word products (Karatsuba on word pairs) and three exact divisions: about 1.8k issue cycles.
__device__ __forceinline__ gf192 gf_mul_small2_over(const gf192 &a, uint32_t y0, uint32_t y1, int k1, int k2)
{
    const holes4 ys0 = holes_split(y0), ys1 = holes_split(y1), ysm = holes_add(ys0, ys1);
    uint32_t c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const holes4 a0 = holes_split(a.w[2 * i]), a1 = holes_split(a.w[2 * i + 1]);
        uint32_t l0, h0, l1, h1, lm, hm;
        clmul32_holes(a0, ys0, l0, h0);
        clmul32_holes(a1, ys1, l1, h1);
        clmul32_holes(holes_add(a0, a1), ysm, lm, hm);
        // (a0 + a1 X)(y0 + y1 X) = d0 + X (m + d0 + d1) + X^2 d1
        c[2 * i] ^= l0;
        c[2 * i + 1] ^= h0 ^ xor3(lm, l0, l1);
        c[2 * i + 2] ^= l1 ^ xor3(hm, h0, h1);
        c[2 * i + 3] ^= h1;
    }
    // bits 192..255 fold back through x^192 = 1 + x + x^2 + x^7
    const uint32_t h0 = c[6], h1 = c[7];
    gf192 v;
    v.w[0] = c[0] ^ h0 ^ (h0 << 1) ^ (h0 << 2) ^ (h0 << 7);
    v.w[1] = c[1] ^ h1 ^ xor3(gf_funnel(h1, h0, 1), gf_funnel(h1, h0, 2), gf_funnel(h1, h0, 7));
    v.w[2] = c[2] ^ (h1 >> 31) ^ (h1 >> 30) ^ (h1 >> 25);
    v.w[3] = c[3]; v.w[4] = c[4]; v.w[5] = c[5];
    return gf_div_1px(gf_div_xk(gf_div_xk(v, k1), k2));
}

// ---------------------------------------------------------------------------------------------------
// Product by a WAVE-UNIFORM multiplier c (every active lane of the wavefront holds the same c): the
// carry-less product is one hand-written asm block (iopx/gfx950_comb.h, generated by tools/gen_comb_asm.py): a 4-bit-window comb whose
// window value, being wave-uniform, selects a code block by a scalar jump instead of a table entry by relative addressing — about 0.45k
// VALU ops, most of them fast-class, against ~1.0k for gf_mul (1.6k against 3.4k issue cycles).  The caller guarantees uniformity of c.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ gf192 gf_mul_uniform(const gf192 &a, const gf192 &c_uniform)
{
    uint32_t c[6], r[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_readfirstlane(c_uniform.w[i]);
    comb_clmul_192_uniform(r, a.w, c);
    return gf_reduce(r);
}

// Product by a multiplier that is uniform within each HALF of the wavefront (lanes 0..31: c_lo, lanes 32..63: c_hi; both must be wave-uniform
// values, e.g. fetched through the scalar unit from indices computed on readfirstlane'd quantities).  `lane` = the caller's lane within its wavefront:
// unused on the GPU (EXEC selects the half), it tells the CPU emulation — one lane at a time — which half it is in.
__device__ __forceinline__ gf192 gf_mul_halves(const gf192 &a, const gf192 &c_lo, const gf192 &c_hi, int lane)
{
    uint32_t c[6], d[6], r[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) { c[i] = __builtin_amdgcn_readfirstlane(c_lo.w[i]); d[i] = __builtin_amdgcn_readfirstlane(c_hi.w[i]); }
    iopx_set_emu_lane(lane);
    comb_clmul_192_halves(r, a.w, c, d);
    return gf_reduce(r);
}

// ---- subset sums over an affine subspace --------------------------------------------------------------------------------
// v_j = t[0] + sum_{bit k of j} t[1 + k] for the j-th element of a 2^m-point subspace (t[0] the shift term, t[1 + k] the image of
// basis vector k).  A workgroup sweeps 256 consecutive positions, so index bits 0..7 vary across lanes (masked XORs) and the rest
// are uniform: their contribution is looked up in two 512-entry tables of pre-summed entries (bits 8..16 and 17..25) that the
// host appends to the table (gf192_host.h subset_table_with_ext) — two loads from a wave-uniform address instead of up to 18
// dependent conditional ones, which left these kernels waiting on memory (65 % of wave cycles parked in k_fz_add).
#define SUBSET_EXT_ENTRIES 1024
__device__ __forceinline__ gf192 subset_sum_ext(const uint64_t *__restrict__ t, int m, uint32_t jlo, uint32_t jhi_uniform)
{
    gf192 v = gf_load(t, 0);
    const int lo_bits = m < 8 ? m : 8;
    for (int k = 0; k < lo_bits; ++k) {
        const uint32_t mask = 0u - ((jlo >> k) & 1u);
        const gf192 b = gf_load(t, 1 + k);
#pragma unroll
        for (int w = 0; w < 6; ++w) v.w[w] = xor_and(v.w[w], mask, b.w[w]);
    }
    const uint64_t *ext = t + 3 * (size_t)(m + 1);
    if (m > 8) gf_add_to(v, gf_load(ext, jhi_uniform & 511u));
    if (m > 17) gf_add_to(v, gf_load(ext, 512u + ((jhi_uniform >> 9) & 511u)));
    for (int k = 26; k < m; ++k) if ((jhi_uniform >> (k - 8)) & 1u) gf_add_to(v, gf_load(t, 1 + k));
    return v;
}

// ---- squaring and inversion -------------------------------------------------------------------------------------------
// Squaring is GF(2)-linear: bit i moves to bit 2i (four shift-and-mask steps per 16 bits), then the usual reduction: about a
// sixth of a product.  Inversion is a^(2^192 - 2) by the Itoh-Tsujii chain on 191 = 0b10111111: 191 squarings + 10 products.
__device__ __forceinline__ uint32_t gf_spread16(uint32_t x)
{
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__device__ __forceinline__ gf192 gf_sqr(const gf192 &a)
{
    uint32_t c[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        c[2 * i] = gf_spread16(a.w[i] & 0xFFFFu);
        c[2 * i + 1] = gf_spread16(a.w[i] >> 16);
    }
    return gf_reduce(c);
}

__device__ inline gf192 gf_inv(const gf192 &a)
{
    gf192 beta = a;                         // beta_k = a^(2^k - 1)
    int k = 1;
    for (int bit = 6; bit >= 0; --bit) {
        gf192 t = beta;
        for (int i = 0; i < k; ++i) t = gf_sqr(t);
        beta = gf_mul(t, beta);             // beta_2k
        k *= 2;
        if ((191 >> bit) & 1) { beta = gf_mul(gf_sqr(beta), a); k += 1; }
    }
    return gf_sqr(beta);
}

