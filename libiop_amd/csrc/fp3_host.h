// Host-side arithmetic in F_p (edwards_Fr, Montgomery words) for the O(log n) per-call constants of the
// multiplicative-domain kernels (inverse shift, challenge powers, n^-1).  Never touches codeword-sized data.
#pragma once
#include <cstdint>
#include <cstring>

namespace iopx {

struct hfp3 {
    uint64_t w[3];
    typedef unsigned __int128 u128;
    static constexpr uint64_t P[3] = { 0x1de5532780000001ull, 0xc4e2e493b92e12ccull, 0x0010357f274a8e56ull };
    static constexpr uint64_t INV = 0xdde553277fffffffull;

    hfp3() { w[0] = w[1] = w[2] = 0; }
    static hfp3 from_words(const uint64_t *p) { hfp3 r; r.w[0] = p[0]; r.w[1] = p[1]; r.w[2] = p[2]; return r; }
    bool operator==(const hfp3 &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2]; }
    bool is_zero() const { return (w[0] | w[1] | w[2]) == 0; }

    static bool geq_p(const uint64_t *a)
    {
        for (int i = 2; i >= 0; --i) if (a[i] != P[i]) return a[i] > P[i];
        return true;
    }
    static void sub_p(uint64_t *a)
    {
        u128 borrow = 0;
        for (int i = 0; i < 3; ++i) { const u128 d = (u128)a[i] - P[i] - borrow; a[i] = (uint64_t)d; borrow = (d >> 64) & 1; }
    }
    hfp3 operator*(const hfp3 &b) const
    {
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 3; ++i) {
            u128 carry = 0;
            for (int j = 0; j < 3; ++j) { const u128 cur = (u128)w[j] * b.w[i] + t[j] + carry; t[j] = (uint64_t)cur; carry = cur >> 64; }
            u128 cur = (u128)t[3] + carry;
            t[3] = (uint64_t)cur; t[4] = (uint64_t)(cur >> 64);
            const uint64_t m = t[0] * INV;
            cur = (u128)m * P[0] + t[0];
            carry = cur >> 64;
            for (int j = 1; j < 3; ++j) { cur = (u128)m * P[j] + t[j] + carry; t[j - 1] = (uint64_t)cur; carry = cur >> 64; }
            cur = (u128)t[3] + carry;
            t[2] = (uint64_t)cur;
            t[3] = t[4] + (uint64_t)(cur >> 64);
        }
        hfp3 r; r.w[0] = t[0]; r.w[1] = t[1]; r.w[2] = t[2];
        if (t[3] || geq_p(r.w)) sub_p(r.w);
        return r;
    }
    hfp3 operator-(const hfp3 &b) const
    {
        hfp3 r;
        u128 borrow = 0;
        for (int i = 0; i < 3; ++i) { const u128 d = (u128)w[i] - b.w[i] - borrow; r.w[i] = (uint64_t)d; borrow = (d >> 64) & 1; }
        if (borrow) { u128 carry = 0; for (int i = 0; i < 3; ++i) { const u128 t = (u128)r.w[i] + P[i] + carry; r.w[i] = (uint64_t)t; carry = t >> 64; } }
        return r;
    }
    hfp3 squared() const { return *this * *this; }
    static hfp3 one()           // R mod p, by doubling 1 192 times
    {
        hfp3 r; r.w[0] = 1;
        for (int i = 0; i < 192; ++i) {
            uint64_t carry = 0;
            for (int k = 0; k < 3; ++k) { const uint64_t nc = r.w[k] >> 63; r.w[k] = (r.w[k] << 1) | carry; carry = nc; }
            if (carry || geq_p(r.w)) sub_p(r.w);
        }
        return r;
    }
    hfp3 pow_limbs(const uint64_t *e, int n) const
    {
        hfp3 r = one();
        for (int i = 64 * n - 1; i >= 0; --i) { r = r.squared(); if ((e[i / 64] >> (i % 64)) & 1) r = r * *this; }
        return r;
    }
    hfp3 pow(uint64_t e) const { return pow_limbs(&e, 1); }
    hfp3 inverse() const { uint64_t e[3] = { P[0] - 2, P[1], P[2] }; return pow_limbs(e, 3); }
    static hfp3 from_uint(uint64_t v)      // v * R mod p = v (raw) * R^2 * R^-1
    {
        hfp3 raw; raw.w[0] = v;
        hfp3 r2 = one();                    // R mod p -> R^2 mod p by 192 more doublings
        for (int i = 0; i < 192; ++i) {
            uint64_t carry = 0;
            for (int k = 0; k < 3; ++k) { const uint64_t nc = r2.w[k] >> 63; r2.w[k] = (r2.w[k] << 1) | carry; carry = nc; }
            if (carry || geq_p(r2.w)) sub_p(r2.w);
        }
        return raw * r2;
    }
    // The device multiplies in Montgomery radix 2^203 (fp3_dev.h): a multiplier t is uploaded as t * 2^203 mod p, i.e. the
    // stored words of t * 2^11.
    hfp3 table_form() const { return *this * from_uint(2048); }
};

} // namespace iopx
