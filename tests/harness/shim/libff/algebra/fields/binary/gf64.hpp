// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libff/algebra/fields/binary/gf64.hpp — GF(2)[x] / (x^64 + x^4 + x^3 + x + 1), one 64-bit word,
// libff's published interface.  libiop's own test files run most of their additive cases over this field (no kernel of this repository serves it: those
// cases exercise the reference's generic code and this shim).
#pragma once
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <vector>
#include <libff/algebra/field_utils/field_utils.hpp>

namespace libff {

class gf64 {
public:
    static const constexpr uint64_t modulus_ = 0b11011;
    static const constexpr uint64_t num_bits = 64;

    explicit gf64() : value_(0) {}
    explicit gf64(const uint64_t value) : value_(value) {}

    gf64 &operator+=(const gf64 &o) { value_ ^= o.value_; return *this; }
    gf64 &operator-=(const gf64 &o) { value_ ^= o.value_; return *this; }
    gf64 &operator*=(const gf64 &o)
    {
        uint64_t lo = 0, hi = 0;                         // carry-less 64 x 64 -> 128, bit by bit
        for (int i = 0; i < 64; ++i)
            if ((o.value_ >> i) & 1) { lo ^= value_ << i; if (i) hi ^= value_ >> (64 - i); }
        for (int pass = 0; pass < 2; ++pass) {           // x^64 = x^4 + x^3 + x + 1
            const uint64_t t = hi;
            hi = (t >> 60) ^ (t >> 61) ^ (t >> 63);
            lo ^= t ^ (t << 1) ^ (t << 3) ^ (t << 4);
        }
        value_ = lo;
        return *this;
    }
    gf64 &operator^=(const unsigned long pow) { *this = power<gf64>(*this, pow); return *this; }
    gf64 &square() { return *this *= gf64(*this); }
    gf64 &invert() { *this = inverse(); return *this; }
    gf64 operator+(const gf64 &o) const { gf64 r(*this); return r += o; }
    gf64 operator-(const gf64 &o) const { gf64 r(*this); return r -= o; }
    gf64 operator-() const { return *this; }
    gf64 operator*(const gf64 &o) const { gf64 r(*this); return r *= o; }
    gf64 operator^(const unsigned long pow) const { return power<gf64>(*this, pow); }
    template<mp_size_t m> gf64 operator^(const bigint<m> &pow) const { return power<gf64, m>(*this, pow); }
    gf64 squared() const { gf64 r(*this); return r.square(); }
    gf64 inverse() const                                 // a^(2^64 - 2)
    {
        gf64 r = one(), sq = squared();
        for (int i = 1; i < 64; ++i) { r *= sq; sq.square(); }
        return r;
    }
    void randomize() { *this = random_element(); }
    void clear() { value_ = 0; }
    bool operator==(const gf64 &o) const { return value_ == o.value_; }
    bool operator!=(const gf64 &o) const { return value_ != o.value_; }
    bool is_zero() const { return value_ == 0; }
    void print() const { printf("%016lx\n", (unsigned long)value_); }
    std::vector<uint64_t> to_words() const { return std::vector<uint64_t>({ value_ }); }
    bool from_words(std::vector<uint64_t> words) { value_ = words[0]; return true; }

    static uint64_t &stream_state() { static uint64_t s = 0x6a09e667f3bcc908ull; return s; }
    static void seed_random(const uint64_t seed) { stream_state() = seed; }
    static gf64 random_element()                         // SplitMix64
    {
        uint64_t z = (stream_state() += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return gf64(z ^ (z >> 31));
    }
    static gf64 zero() { return gf64(0); }
    static gf64 one() { return gf64(1); }
    static gf64 multiplicative_generator;                // = gf64(2)
    static std::size_t ceil_size_in_bits() { return num_bits; }
    static std::size_t floor_size_in_bits() { return num_bits; }
    static constexpr std::size_t extension_degree() { return 64; }

    friend std::ostream &operator<<(std::ostream &out, const gf64 &el) { return out << el.value_; }
    friend std::istream &operator>>(std::istream &in, gf64 &el) { return in >> el.value_; }

private:
    uint64_t value_;
};

template<> struct is_additive<gf64> { static const bool value = true; };

} // namespace libff
