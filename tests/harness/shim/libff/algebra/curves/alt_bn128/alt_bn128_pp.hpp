// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libff's alt_bn128_pp.hpp — only the scalar field alt_bn128_Fr, with the interface of libff's
// Fp_model<4, alt_bn128_modulus_r> that libiop names: four little-endian 64-bit words of the Montgomery representation, R = 2^256,
// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (254 bits), 2-adicity 28, generator 5 (libff's published constants,
// recalled).  libiop wires Poseidon for this field only; its test files use it for the algebraic-hash variants.  Self-contained arithmetic (no kernel of this
// repository is bound to this field through the stubs).
#pragma once
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <vector>
#include <libff/algebra/field_utils/field_utils.hpp>

namespace libff {

class alt_bn128_Fr {
public:
    typedef unsigned __int128 u128;
    static const mp_size_t num_limbs = 4;
    static const std::size_t num_bits = 254;
    static const std::size_t s = 28;
    static bigint<4> mod;
    static alt_bn128_Fr multiplicative_generator, root_of_unity;

    bigint<4> mont_repr;

    alt_bn128_Fr() {}
    alt_bn128_Fr(const bigint<4> &b) { mont_repr = mul(b, R2()); }
    alt_bn128_Fr(const long x, const bool is_unsigned = false)
    {
        if (x >= 0 || is_unsigned) mont_repr = mul(bigint<4>((unsigned long)x), R2());
        else { alt_bn128_Fr p(bigint<4>((unsigned long)(-x))); *this = -p; }
    }

    alt_bn128_Fr &operator+=(const alt_bn128_Fr &o)
    {
        u128 carry = 0;
        for (int i = 0; i < 4; ++i) { const u128 v = (u128)mont_repr.data[i] + o.mont_repr.data[i] + carry; mont_repr.data[i] = (uint64_t)v; carry = v >> 64; }
        if (carry || geq_p(mont_repr)) sub_p(mont_repr);
        return *this;
    }
    alt_bn128_Fr &operator-=(const alt_bn128_Fr &o)
    {
        u128 borrow = 0;
        for (int i = 0; i < 4; ++i) { const u128 d = (u128)mont_repr.data[i] - o.mont_repr.data[i] - borrow; mont_repr.data[i] = (uint64_t)d; borrow = (d >> 64) & 1; }
        if (borrow) { u128 carry = 0; for (int i = 0; i < 4; ++i) { const u128 v = (u128)mont_repr.data[i] + P()[i] + carry; mont_repr.data[i] = (uint64_t)v; carry = v >> 64; } }
        return *this;
    }
    alt_bn128_Fr &operator*=(const alt_bn128_Fr &o) { mont_repr = mul(mont_repr, o.mont_repr); return *this; }
    alt_bn128_Fr &operator^=(const unsigned long pow) { *this = power<alt_bn128_Fr>(*this, pow); return *this; }
    template<mp_size_t m> alt_bn128_Fr &operator^=(const bigint<m> &pow) { *this = power<alt_bn128_Fr, m>(*this, pow); return *this; }
    alt_bn128_Fr operator+(const alt_bn128_Fr &o) const { alt_bn128_Fr r(*this); return r += o; }
    alt_bn128_Fr operator-(const alt_bn128_Fr &o) const { alt_bn128_Fr r(*this); return r -= o; }
    alt_bn128_Fr operator*(const alt_bn128_Fr &o) const { alt_bn128_Fr r(*this); return r *= o; }
    alt_bn128_Fr operator-() const { alt_bn128_Fr z; return z -= *this; }
    alt_bn128_Fr operator^(const unsigned long pow) const { return power<alt_bn128_Fr>(*this, pow); }
    template<mp_size_t m> alt_bn128_Fr operator^(const bigint<m> &pow) const { return power<alt_bn128_Fr, m>(*this, pow); }
    alt_bn128_Fr squared() const { return *this * *this; }
    alt_bn128_Fr &square() { *this = squared(); return *this; }
    alt_bn128_Fr inverse() const { bigint<4> e = mod; e.data[0] -= 2; return power<alt_bn128_Fr, 4>(*this, e); }
    alt_bn128_Fr &invert() { *this = inverse(); return *this; }

    bool operator==(const alt_bn128_Fr &o) const { return mont_repr == o.mont_repr; }
    bool operator!=(const alt_bn128_Fr &o) const { return !(*this == o); }
    bool is_zero() const { return mont_repr.is_zero(); }
    void clear() { mont_repr.clear(); }
    void print() const { const bigint<4> b = as_bigint(); printf("%016lx%016lx%016lx%016lx\n", (unsigned long)b.data[3], (unsigned long)b.data[2], (unsigned long)b.data[1], (unsigned long)b.data[0]); }
    void randomize() { *this = random_element(); }
    bigint<4> as_bigint() const { return mul(mont_repr, bigint<4>(1ul)); }
    unsigned long as_ulong() const { return as_bigint().as_ulong(); }
    std::vector<uint64_t> to_words() const { const bigint<4> b = as_bigint(); return std::vector<uint64_t>(b.data, b.data + 4); }

    static uint64_t &stream_state() { static uint64_t v = 0xbb67ae8584caa73bull; return v; }
    static void seed_random(const uint64_t seed) { stream_state() = seed; }
    static alt_bn128_Fr random_element()
    {
        bigint<4> c;
        for (int k = 0; k < 4; ++k) {
            uint64_t z = (stream_state() += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            c.data[k] = z ^ (z >> 31);
        }
        alt_bn128_Fr r;
        r.mont_repr = mul(c, R2());                  // the Montgomery product with R^2 reduces any 256-bit value
        return r;
    }
    static alt_bn128_Fr zero() { return alt_bn128_Fr(); }
    static alt_bn128_Fr one() { return alt_bn128_Fr(1); }
    static std::size_t size_in_bits() { return num_bits; }
    static std::size_t ceil_size_in_bits() { return num_bits; }
    static std::size_t floor_size_in_bits() { return num_bits - 1; }
    static constexpr std::size_t extension_degree() { return 1; }
    static alt_bn128_Fr get_root_of_unity(const std::size_t n)
    {
        const std::size_t logn = libff::log2(n);
        if (n != ((std::size_t)1 << logn)) throw std::invalid_argument("libff::get_root_of_unity: expected n == (1u << logn)");
        if (logn > s) throw std::invalid_argument("libff::get_root_of_unity: expected logn <= FieldT::s");
        alt_bn128_Fr omega = root_of_unity;
        for (std::size_t i = s; i > logn; --i) omega *= omega;
        return omega;
    }
    friend std::ostream &operator<<(std::ostream &out, const alt_bn128_Fr &el) { const bigint<4> b = el.as_bigint(); return out << b.data[0] << " " << b.data[1] << " " << b.data[2] << " " << b.data[3]; }
    friend std::istream &operator>>(std::istream &in, alt_bn128_Fr &el) { bigint<4> b; in >> b.data[0] >> b.data[1] >> b.data[2] >> b.data[3]; el = alt_bn128_Fr(b); return in; }

    static const uint64_t *P()
    {
        static const uint64_t p[4] = { 0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull };
        return p;
    }
private:
    static bool geq_p(const bigint<4> &a) { for (int i = 3; i >= 0; --i) if (a.data[i] != P()[i]) return a.data[i] > P()[i]; return true; }
    static void sub_p(bigint<4> &a) { u128 borrow = 0; for (int i = 0; i < 4; ++i) { const u128 d = (u128)a.data[i] - P()[i] - borrow; a.data[i] = (uint64_t)d; borrow = (d >> 64) & 1; } }
    static uint64_t INV()                            // -p^-1 mod 2^64 by Newton iteration
    {
        static const uint64_t inv = [] { uint64_t x = 1; for (int i = 0; i < 7; ++i) x *= 2 - P()[0] * x; return (uint64_t)(0 - x); }();
        return inv;
    }
    static bigint<4> mul(const bigint<4> &a, const bigint<4> &b)     // a b / R mod p (CIOS)
    {
        uint64_t t[6] = { 0, 0, 0, 0, 0, 0 };
        for (int i = 0; i < 4; ++i) {
            u128 carry = 0;
            for (int j = 0; j < 4; ++j) { const u128 cur = (u128)a.data[j] * b.data[i] + t[j] + carry; t[j] = (uint64_t)cur; carry = cur >> 64; }
            u128 cur = (u128)t[4] + carry;
            t[4] = (uint64_t)cur; t[5] = (uint64_t)(cur >> 64);
            const uint64_t m = t[0] * INV();
            cur = (u128)m * P()[0] + t[0];
            carry = cur >> 64;
            for (int j = 1; j < 4; ++j) { cur = (u128)m * P()[j] + t[j] + carry; t[j - 1] = (uint64_t)cur; carry = cur >> 64; }
            cur = (u128)t[4] + carry;
            t[3] = (uint64_t)cur;
            t[4] = t[5] + (uint64_t)(cur >> 64);
        }
        bigint<4> r;
        for (int i = 0; i < 4; ++i) r.data[i] = t[i];
        if (t[4] || geq_p(r)) sub_p(r);
        return r;
    }
    static const bigint<4> &R2()                     // R^2 mod p: 1 doubled 512 times
    {
        static const bigint<4> r2 = [] {
            bigint<4> r(1ul);
            for (int i = 0; i < 512; ++i) {
                uint64_t carry = 0;
                for (int k = 0; k < 4; ++k) { const uint64_t nc = r.data[k] >> 63; r.data[k] = (r.data[k] << 1) | carry; carry = nc; }
                if (carry || geq_p(r)) sub_p(r);
            }
            return r;
        }();
        return r2;
    }
};

template<> struct is_multiplicative<alt_bn128_Fr> { static const bool value = true; };

struct alt_bn128_pp {
    static void init_public_params()
    {
        for (int i = 0; i < 4; ++i) alt_bn128_Fr::mod.data[i] = alt_bn128_Fr::P()[i];
        alt_bn128_Fr::multiplicative_generator = alt_bn128_Fr(5);
        bigint<4> e, pm1 = alt_bn128_Fr::mod;        // (p - 1) / 2^28
        pm1.data[0] -= 1;
        for (int i = 0; i < 4; ++i) e.data[i] = (pm1.data[i] >> 28) | (i < 3 ? pm1.data[i + 1] << 36 : 0);
        alt_bn128_Fr::root_of_unity = alt_bn128_Fr::multiplicative_generator ^ e;
    }
};

} // namespace libff
