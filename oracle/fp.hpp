// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// Prime-field arithmetic restated from the published definition of libff's Fp_model (scipr-lab/libff,
// absent submodule of the reference, commit unpinned): an element is its Montgomery representative
// a * R mod p with R = 2^(64 * limbs), stored as little-endian 64-bit limbs (`mont_repr`).  libiop hashes
// and samples these limbs raw (libiop/bcs/hashing/blake2b.tcc:148-152, :197-227), so device buffers keep
// exactly this representation.
//   edwards_Fr : p = 1552511030102430251236801561344621993261920897571225601 (181 bits, 3 limbs),
//                2-adicity 31, multiplicative_generator 19,
//                root_of_unity = 19^((p-1)/2^31) = 695314865466598274460565335217615316274564719601897184
//                (recomputed in tests/test_oracle_fp.py; p's primality and 2-adicity are checked there too).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace oracle {

typedef unsigned __int128 u128;

struct edwards_Fr_params {
    static constexpr int limbs = 3;
    static constexpr uint64_t modulus[3] = { 0x1de5532780000001ull, 0xc4e2e493b92e12ccull, 0x0010357f274a8e56ull };
    static constexpr uint64_t inv = 0xdde553277fffffffull;      // -p^{-1} mod 2^64
    static constexpr uint64_t generator = 19;
    static constexpr int two_adicity = 31;
};

// alt_bn128 Fr: p = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (254 bits, 4 limbs),
// 2-adicity 28, multiplicative_generator 5 — the only field the reference wires Poseidon for.
struct alt_bn128_Fr_params {
    static constexpr int limbs = 4;
    static constexpr uint64_t modulus[4] = { 0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull };
    static constexpr uint64_t inv = 0xc2e1f593efffffffull;
    static constexpr uint64_t generator = 5;
    static constexpr int two_adicity = 28;
};

template<typename P>
struct Fp {
    static constexpr int N = P::limbs;
    uint64_t mont[N];

    Fp() { for (int i = 0; i < N; ++i) mont[i] = 0; }
    explicit Fp(uint64_t v) { uint64_t c[N] = {0}; c[0] = v; *this = from_canonical(c); }

    static bool geq_mod(const uint64_t *a)
    {
        for (int i = N - 1; i >= 0; --i) { if (a[i] != P::modulus[i]) return a[i] > P::modulus[i]; }
        return true;
    }
    static void sub_mod(uint64_t *a)
    {
        u128 borrow = 0;
        for (int i = 0; i < N; ++i) {
            const u128 d = (u128)a[i] - P::modulus[i] - borrow;
            a[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
    }

    static Fp zero() { return Fp(); }
    static Fp one() { return Fp(1); }
    static Fp multiplicative_generator() { return Fp(P::generator); }

    // R^2 mod p by repeated doubling (2 * 64 N doublings of 1)
    static Fp R2()
    {
        Fp r;
        r.mont[0] = 1;
        for (int i = 0; i < 2 * 64 * N; ++i) {
            uint64_t carry = 0;
            for (int k = 0; k < N; ++k) { const uint64_t nc = r.mont[k] >> 63; r.mont[k] = (r.mont[k] << 1) | carry; carry = nc; }
            if (carry || geq_mod(r.mont)) sub_mod(r.mont);
        }
        return r;
    }

    static Fp from_canonical(const uint64_t *c)
    {
        Fp x;
        for (int i = 0; i < N; ++i) x.mont[i] = c[i];
        static const Fp r2 = R2();
        return mont_mul(x, r2);
    }
    void to_canonical(uint64_t *c) const
    {
        Fp one_raw;
        one_raw.mont[0] = 1;
        const Fp r = mont_mul(*this, one_raw);
        for (int i = 0; i < N; ++i) c[i] = r.mont[i];
    }

    // CIOS Montgomery product: a * b * R^{-1} mod p
    static Fp mont_mul(const Fp &a, const Fp &b)
    {
        uint64_t t[N + 2];
        for (int i = 0; i < N + 2; ++i) t[i] = 0;
        for (int i = 0; i < N; ++i) {
            u128 carry = 0;
            for (int j = 0; j < N; ++j) {
                const u128 cur = (u128)a.mont[j] * b.mont[i] + t[j] + carry;
                t[j] = (uint64_t)cur;
                carry = cur >> 64;
            }
            u128 cur = (u128)t[N] + carry;
            t[N] = (uint64_t)cur;
            t[N + 1] = (uint64_t)(cur >> 64);
            const uint64_t m = t[0] * P::inv;
            cur = (u128)m * P::modulus[0] + t[0];
            carry = cur >> 64;
            for (int j = 1; j < N; ++j) {
                cur = (u128)m * P::modulus[j] + t[j] + carry;
                t[j - 1] = (uint64_t)cur;
                carry = cur >> 64;
            }
            cur = (u128)t[N] + carry;
            t[N - 1] = (uint64_t)cur;
            t[N] = t[N + 1] + (uint64_t)(cur >> 64);
        }
        Fp r;
        for (int i = 0; i < N; ++i) r.mont[i] = t[i];
        if (t[N] || geq_mod(r.mont)) sub_mod(r.mont);
        return r;
    }

    bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; ++i) a |= mont[i]; return a == 0; }
    bool operator==(const Fp &o) const { uint64_t a = 0; for (int i = 0; i < N; ++i) a |= mont[i] ^ o.mont[i]; return a == 0; }
    bool operator!=(const Fp &o) const { return !(*this == o); }

    Fp &operator+=(const Fp &o)
    {
        u128 carry = 0;
        for (int i = 0; i < N; ++i) { const u128 s = (u128)mont[i] + o.mont[i] + carry; mont[i] = (uint64_t)s; carry = s >> 64; }
        if (carry || geq_mod(mont)) sub_mod(mont);
        return *this;
    }
    Fp &operator-=(const Fp &o)
    {
        u128 borrow = 0;
        for (int i = 0; i < N; ++i) { const u128 d = (u128)mont[i] - o.mont[i] - borrow; mont[i] = (uint64_t)d; borrow = (d >> 64) & 1; }
        if (borrow) {
            u128 carry = 0;
            for (int i = 0; i < N; ++i) { const u128 s = (u128)mont[i] + P::modulus[i] + carry; mont[i] = (uint64_t)s; carry = s >> 64; }
        }
        return *this;
    }
    Fp &operator*=(const Fp &o) { *this = mont_mul(*this, o); return *this; }
    Fp operator+(const Fp &o) const { Fp r(*this); r += o; return r; }
    Fp operator-(const Fp &o) const { Fp r(*this); r -= o; return r; }
    Fp operator*(const Fp &o) const { return mont_mul(*this, o); }
    Fp operator-() const { Fp r; r -= *this; return r; }
    Fp squared() const { return mont_mul(*this, *this); }

    // exponent given as little-endian limbs
    Fp pow_limbs(const uint64_t *e, int n) const
    {
        Fp r = one();
        for (int i = 64 * n - 1; i >= 0; --i) {
            r = r.squared();
            if ((e[i / 64] >> (i % 64)) & 1) r *= *this;
        }
        return r;
    }
    Fp pow(uint64_t e) const { return pow_limbs(&e, 1); }

    Fp inverse() const      // a^(p-2)
    {
        uint64_t e[N];
        for (int i = 0; i < N; ++i) e[i] = P::modulus[i];
        e[0] -= 2;          // p is odd and > 2: no borrow
        return pow_limbs(e, N);
    }

    // multiplicative_generator^((p-1)/order) for a power-of-two order (subgroup.tcc:55-59)
    static Fp subgroup_generator(size_t order)
    {
        uint64_t e[N];
        for (int i = 0; i < N; ++i) e[i] = P::modulus[i];
        e[0] -= 1;
        size_t lg = 0;
        while (((size_t)1 << lg) < order) ++lg;
        for (size_t s = 0; s < lg; ++s) {      // e >>= 1
            for (int i = 0; i < N; ++i) e[i] = (e[i] >> 1) | (i + 1 < N ? e[i + 1] << 63 : 0);
        }
        return multiplicative_generator().pow_limbs(e, N);
    }
};

typedef Fp<edwards_Fr_params> edwards_Fr;
typedef Fp<alt_bn128_Fr_params> alt_bn128_Fr;

} // namespace oracle
