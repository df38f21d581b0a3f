// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libff/algebra/fields/binary/gf192.hpp with libff's published interface and
// layout (three little-endian 64-bit words, GF(2)[x] / (x^192 + x^7 + x^2 + x + 1)); the arithmetic is this repository's own host code
// (libiop_amd/csrc/gf192_host.h).  It pins nothing about libff's bytes: it lets libiop's protocol code run over the layout the kernels assume.
// random_element() is a seeded SplitMix64 stream (element i = words 3i, 3i+1, 3i+2 of the stream), so that the reference's own
// generate_r1cs_example builds the instance the parity tests use (oracle/aurora.hpp seeded_element).
#pragma once
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <vector>
#include <libff/algebra/field_utils/field_utils.hpp>
#include "gf192_host.h"

namespace libff {

class gf192 {
public:
    static const constexpr uint64_t modulus_ = 0b10000111;
    static const constexpr uint64_t num_bits = 192;

    explicit gf192() { value_[0] = value_[1] = value_[2] = 0; }
    explicit gf192(const uint64_t value_low) { value_[0] = value_low; value_[1] = value_[2] = 0; }
    explicit gf192(const uint64_t value_high, const uint64_t value_mid, const uint64_t value_low) { value_[0] = value_low; value_[1] = value_mid; value_[2] = value_high; }

    gf192 &operator+=(const gf192 &o) { for (int i = 0; i < 3; ++i) value_[i] ^= o.value_[i]; return *this; }
    gf192 &operator-=(const gf192 &o) { return *this += o; }
    gf192 &operator*=(const gf192 &o) { set(h() * o.h()); return *this; }
    gf192 &operator^=(const unsigned long pow) { *this = power<gf192>(*this, pow); return *this; }
    gf192 &square() { set(h().squared()); return *this; }
    gf192 &invert() { set(h().inverse()); return *this; }

    gf192 operator+(const gf192 &o) const { gf192 r(*this); return r += o; }
    gf192 operator-(const gf192 &o) const { gf192 r(*this); return r -= o; }
    gf192 operator-() const { return *this; }
    gf192 operator*(const gf192 &o) const { gf192 r(*this); return r *= o; }
    gf192 operator^(const unsigned long pow) const { return power<gf192>(*this, pow); }
    template<mp_size_t m> gf192 operator^(const bigint<m> &pow) const { return power<gf192, m>(*this, pow); }
    gf192 squared() const { gf192 r(*this); return r.square(); }
    gf192 inverse() const { gf192 r(*this); return r.invert(); }

    void randomize() { *this = random_element(); }
    void clear() { value_[0] = value_[1] = value_[2] = 0; }
    bool operator==(const gf192 &o) const { return value_[0] == o.value_[0] && value_[1] == o.value_[1] && value_[2] == o.value_[2]; }
    bool operator!=(const gf192 &o) const { return !(*this == o); }
    bool is_zero() const { return (value_[0] | value_[1] | value_[2]) == 0; }
    void print() const { printf("%016lx%016lx%016lx\n", (unsigned long)value_[2], (unsigned long)value_[1], (unsigned long)value_[0]); }

    std::vector<uint64_t> to_words() const { return std::vector<uint64_t>({ value_[0], value_[1], value_[2] }); }
    bool from_words(std::vector<uint64_t> words) { for (int i = 0; i < 3; ++i) value_[i] = words[i]; return true; }

    static uint64_t &stream_seed() { static uint64_t s = 0; return s; }
    static uint64_t &stream_next() { static uint64_t n = 0; return n; }
    static void seed_random(const uint64_t seed) { stream_seed() = seed; stream_next() = 0; }
    static uint64_t splitmix64_at(uint64_t seed, uint64_t index)
    {
        uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    static gf192 random_element()
    {
        gf192 r;
        const uint64_t i = stream_next()++;
        for (int k = 0; k < 3; ++k) r.value_[k] = splitmix64_at(stream_seed(), 3 * i + k);
        return r;
    }

    static gf192 zero() { return gf192(0); }
    static gf192 one() { return gf192(1); }
    static gf192 multiplicative_generator;          // = gf192(2)

    static std::size_t ceil_size_in_bits() { return num_bits; }
    static std::size_t floor_size_in_bits() { return num_bits; }
    static constexpr std::size_t extension_degree() { return 192; }
    template<mp_size_t n> static constexpr bigint<n> field_char() { return bigint<n>(2); }

    friend std::ostream &operator<<(std::ostream &out, const gf192 &el) { return out << el.value_[0] << " " << el.value_[1] << " " << el.value_[2]; }
    friend std::istream &operator>>(std::istream &in, gf192 &el) { return in >> el.value_[0] >> el.value_[1] >> el.value_[2]; }

private:
    uint64_t value_[3];
    iopx::hgf192 h() const { return iopx::hgf192::from_words(value_); }
    void set(const iopx::hgf192 &v) { value_[0] = v.w[0]; value_[1] = v.w[1]; value_[2] = v.w[2]; }
};

template<> struct is_additive<gf192> { static const bool value = true; };

} // namespace libff
