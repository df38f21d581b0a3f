// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libff's edwards_pp.hpp — only the scalar field edwards_Fr, with the interface of
// libff's Fp_model<3, edwards_modulus_r> that libiop names and libff's published layout (three little-endian 64-bit words of the Montgomery
// representation, R = 2^192, p = 1552511030102430251236801561344621993261920897571225601 (181 bits), 2-adicity 31, generator 19 — recalled,
// SURVEY.md §8c).  The arithmetic is this repository's own host code (libiop_amd/csrc/fp3_host.h).  It pins nothing about libff's bytes.
// random_element() is the seeded stream of oracle/aurora.hpp seeded_element: three SplitMix64 words, reduced mod p, to Montgomery form.
#pragma once
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <vector>
#include <libff/algebra/field_utils/field_utils.hpp>
#include "fp3_host.h"

namespace libff {

class edwards_Fr {
public:
    static const mp_size_t num_limbs = 3;
    static const std::size_t num_bits = 181;
    static const std::size_t s = 31;                 // 2-adicity of p - 1
    static bigint<3> mod;
    static edwards_Fr multiplicative_generator;      // 19
    static edwards_Fr root_of_unity;                 // 19^((p - 1) / 2^31)

    bigint<3> mont_repr;

    edwards_Fr() {}
    edwards_Fr(const bigint<3> &b) { set(raw(b) * R2()); }
    edwards_Fr(const long x, const bool is_unsigned = false)
    {
        if (x >= 0 || is_unsigned) { bigint<3> b((unsigned long)x); set(raw(b) * R2()); }
        else { bigint<3> b((unsigned long)(-x)); const iopx::hfp3 zero; set(zero - raw(b) * R2()); }
    }

    edwards_Fr &operator+=(const edwards_Fr &o)
    {
        unsigned __int128 carry = 0;
        uint64_t t[3];
        for (int i = 0; i < 3; ++i) { const unsigned __int128 v = (unsigned __int128)mont_repr.data[i] + o.mont_repr.data[i] + carry; t[i] = (uint64_t)v; carry = v >> 64; }
        if (carry || iopx::hfp3::geq_p(t)) iopx::hfp3::sub_p(t);
        for (int i = 0; i < 3; ++i) mont_repr.data[i] = t[i];
        return *this;
    }
    edwards_Fr &operator-=(const edwards_Fr &o) { set(h() - o.h()); return *this; }
    edwards_Fr &operator*=(const edwards_Fr &o) { set(h() * o.h()); return *this; }
    edwards_Fr &operator^=(const unsigned long pow) { *this = power<edwards_Fr>(*this, pow); return *this; }
    template<mp_size_t m> edwards_Fr &operator^=(const bigint<m> &pow) { *this = power<edwards_Fr, m>(*this, pow); return *this; }
    edwards_Fr operator+(const edwards_Fr &o) const { edwards_Fr r(*this); return r += o; }
    edwards_Fr operator-(const edwards_Fr &o) const { edwards_Fr r(*this); return r -= o; }
    edwards_Fr operator*(const edwards_Fr &o) const { edwards_Fr r(*this); return r *= o; }
    edwards_Fr operator-() const { edwards_Fr r; const iopx::hfp3 zero; r.set(zero - h()); return r; }
    edwards_Fr operator^(const unsigned long pow) const { return power<edwards_Fr>(*this, pow); }
    template<mp_size_t m> edwards_Fr operator^(const bigint<m> &pow) const { return power<edwards_Fr, m>(*this, pow); }
    edwards_Fr squared() const { return *this * *this; }
    edwards_Fr &square() { *this = squared(); return *this; }
    edwards_Fr inverse() const { edwards_Fr r; r.set(h().inverse()); return r; }
    edwards_Fr &invert() { *this = inverse(); return *this; }

    bool operator==(const edwards_Fr &o) const { return mont_repr == o.mont_repr; }
    bool operator!=(const edwards_Fr &o) const { return !(*this == o); }
    bool is_zero() const { return mont_repr.is_zero(); }
    void clear() { mont_repr.clear(); }
    void print() const { const bigint<3> b = as_bigint(); printf("%016lx%016lx%016lx\n", (unsigned long)b.data[2], (unsigned long)b.data[1], (unsigned long)b.data[0]); }
    void randomize() { *this = random_element(); }

    bigint<3> as_bigint() const                      // out of Montgomery form: the Montgomery product with the raw 1
    {
        iopx::hfp3 one_raw; one_raw.w[0] = 1;
        const iopx::hfp3 v = h() * one_raw;
        bigint<3> b;
        for (int i = 0; i < 3; ++i) b.data[i] = v.w[i];
        return b;
    }
    unsigned long as_ulong() const { return as_bigint().as_ulong(); }
    std::vector<uint64_t> to_words() const { const bigint<3> b = as_bigint(); return std::vector<uint64_t>(b.data, b.data + 3); }

    static uint64_t &stream_seed() { static uint64_t v = 0; return v; }
    static uint64_t &stream_next() { static uint64_t v = 0; return v; }
    static void seed_random(const uint64_t seed) { stream_seed() = seed; stream_next() = 0; }
    static uint64_t splitmix64_at(uint64_t seed, uint64_t index)
    {
        uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    static edwards_Fr random_element()
    {
        const uint64_t i = stream_next()++;
        iopx::hfp3 c;
        for (int k = 0; k < 3; ++k) c.w[k] = splitmix64_at(stream_seed(), 3 * i + k);
        edwards_Fr r;
        r.set(c * R2());                             // the Montgomery product with R^2 reduces any 192-bit value
        return r;
    }

    static edwards_Fr zero() { return edwards_Fr(); }
    static edwards_Fr one() { return edwards_Fr(1); }
    static std::size_t size_in_bits() { return num_bits; }
    static std::size_t ceil_size_in_bits() { return num_bits; }
    static std::size_t floor_size_in_bits() { return num_bits - 1; }
    static constexpr std::size_t extension_degree() { return 1; }
    static edwards_Fr get_root_of_unity(const std::size_t n)         // libff::get_root_of_unity for a prime field
    {
        const std::size_t logn = libff::log2(n);
        if (n != ((std::size_t)1 << logn)) throw std::invalid_argument("libff::get_root_of_unity: expected n == (1u << logn)");
        if (logn > s) throw std::invalid_argument("libff::get_root_of_unity: expected logn <= FieldT::s");
        edwards_Fr omega = root_of_unity;
        for (std::size_t i = s; i > logn; --i) omega *= omega;
        return omega;
    }

    friend std::ostream &operator<<(std::ostream &out, const edwards_Fr &el) { const bigint<3> b = el.as_bigint(); return out << b.data[0] << " " << b.data[1] << " " << b.data[2]; }
    friend std::istream &operator>>(std::istream &in, edwards_Fr &el) { bigint<3> b; in >> b.data[0] >> b.data[1] >> b.data[2]; el = edwards_Fr(b); return in; }

private:
    iopx::hfp3 h() const { iopx::hfp3 v; for (int i = 0; i < 3; ++i) v.w[i] = mont_repr.data[i]; return v; }
    void set(const iopx::hfp3 &v) { for (int i = 0; i < 3; ++i) mont_repr.data[i] = v.w[i]; }
    static iopx::hfp3 raw(const bigint<3> &b) { iopx::hfp3 v; for (int i = 0; i < 3; ++i) v.w[i] = b.data[i]; return v; }
    static iopx::hfp3 R2()                           // R^2 mod p: the Montgomery form of R
    {
        static const iopx::hfp3 r2 = [] {
            iopx::hfp3 r = iopx::hfp3::one();
            for (int i = 0; i < 192; ++i) {
                uint64_t carry = 0;
                for (int k = 0; k < 3; ++k) { const uint64_t nc = r.w[k] >> 63; r.w[k] = (r.w[k] << 1) | carry; carry = nc; }
                if (carry || iopx::hfp3::geq_p(r.w)) iopx::hfp3::sub_p(r.w);
            }
            return r;
        }();
        return r2;
    }
};

template<> struct is_multiplicative<edwards_Fr> { static const bool value = true; };

// the one-time initialisation libiop's programs call before using the field
struct edwards_pp {
    static void init_public_params()
    {
        for (int i = 0; i < 3; ++i) edwards_Fr::mod.data[i] = iopx::hfp3::P[i];
        edwards_Fr::multiplicative_generator = edwards_Fr(19);
        bigint<3> e;                                 // (p - 1) / 2^31
        bigint<3> pm1 = edwards_Fr::mod; pm1.data[0] -= 1;
        for (int i = 0; i < 3; ++i) e.data[i] = (pm1.data[i] >> 31) | (i < 2 ? pm1.data[i + 1] << 33 : 0);
        edwards_Fr::root_of_unity = edwards_Fr::multiplicative_generator ^ e;
    }
};

} // namespace libff
