// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it.
//
// Binary-field arithmetic restated from the *published* definition of libff's binary fields.
// libff (scipr-lab/libff, git submodule depends/libff of the reference, commit UNPINNED and the
// directory EMPTY in /root/reference) is a third-party dependency that is absent here, so these
// constants are restated from libff's public sources, not read from files in the reference tree:
//   gf64  = GF(2)[x]/(x^64  + x^4 + x^3 + x + 1)          (libff/algebra/fields/binary/gf64.hpp)
//   gf128 = GF(2)[x]/(x^128 + x^7 + x^2 + x + 1)          (.../gf128.hpp)
//   gf192 = GF(2)[x]/(x^192 + x^7 + x^2 + x + 1)          (.../gf192.hpp)
//   gf256 = GF(2)[x]/(x^256 + x^10 + x^5 + x^2 + 1)       (.../gf256.hpp)
// Elements are little-endian arrays of 64-bit words in the polynomial basis; FieldT(uint64_t v)
// sets word 0.  The reference uses these through FieldT operators only (e.g.
// libiop/algebra/fft.tcc:62-70, libiop/algebra/field_subset/subspace.tcc:93-108).
// Pinning: irreducibility of every modulus and the field axioms are checked in
// tests/test_oracle_field.py; byte-identity with libff itself is NOT checkable in this image
// ("parity unpinned" for the libff constants).  The protocol logic above the fields IS checked against the reference's
// own code: tests/harness compiles libiop's unmodified sources over a stand-in libff and tests/test_reference_harness.py compares this oracle's
// Aurora / Fractal transcripts with theirs byte for byte (container only; digests: tests/golden/reference_over_shim.json).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <map>
#include <string>

#if defined(__PCLMUL__)
#include <immintrin.h>
#endif

namespace oracle {

// ---- block timers under the reference's own block names (libff::enter_block / leave_block: bcs/bcs_prover.tcc:26,43,52,
// protocols/ldt/fri/fri_ldt.tcc:519, algebra/fft.tcc:210,222,382,394, snark/aurora_snark.tcc:126,138): inclusive wall-clock seconds and call
// counts per name, read by bench.py's cpu_baseline leg through oracle_block_times.  Off the hot loops: one clock read at each end of a block. ----
struct block_times {
    static std::map<std::string, std::pair<double, size_t>> &table() { static std::map<std::string, std::pair<double, size_t>> t; return t; }
};
struct timed_block {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit timed_block(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~timed_block()
    {
        auto &e = block_times::table()[name];
        e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        e.second += 1;
    }
};

// ---- carry-less 64x64 -> 128 -------------------------------------------------------------
static inline void clmul64_portable(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi)
{
    // shift-and-xor, one bit of b at a time (reference-free textbook definition)
    uint64_t l = 0, h = 0;
    for (int i = 0; i < 64; ++i) {
        const uint64_t m = (uint64_t)0 - ((b >> i) & 1);
        l ^= (a << i) & m;
        if (i) h ^= (a >> (64 - i)) & m;
    }
    lo = l; hi = h;
}

static inline void clmul64(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi)
{
#if defined(__PCLMUL__)
    const __m128i x = _mm_set_epi64x(0, (long long)a);
    const __m128i y = _mm_set_epi64x(0, (long long)b);
    const __m128i r = _mm_clmulepi64_si128(x, y, 0x00);
    lo = (uint64_t)_mm_cvtsi128_si64(r);
    hi = (uint64_t)_mm_extract_epi64(r, 1);
#else
    clmul64_portable(a, b, lo, hi);
#endif
}

// ---- generic binary field with W 64-bit words and a low-weight modulus tail ----------------
// x^(64W) = TAIL(x) where TAIL has degree < 64 - (its own degree) so one folding step per word
// suffices twice (standard pentanomial reduction).
template<int W, uint64_t TAIL>
struct gf2n {
    uint64_t w[W];

    static constexpr int num_words = W;
    static constexpr int extension_degree = 64 * W;

    gf2n() { for (int i = 0; i < W; ++i) w[i] = 0; }
    explicit gf2n(uint64_t v) { w[0] = v; for (int i = 1; i < W; ++i) w[i] = 0; }

    static gf2n zero() { return gf2n(); }
    static gf2n one() { return gf2n(1); }
    static gf2n multiplicative_generator() { return gf2n(2); }

    bool is_zero() const { uint64_t a = 0; for (int i = 0; i < W; ++i) a |= w[i]; return a == 0; }
    bool operator==(const gf2n &o) const { uint64_t a = 0; for (int i = 0; i < W; ++i) a |= w[i] ^ o.w[i]; return a == 0; }
    bool operator!=(const gf2n &o) const { return !(*this == o); }

    gf2n &operator+=(const gf2n &o) { for (int i = 0; i < W; ++i) w[i] ^= o.w[i]; return *this; }
    gf2n &operator-=(const gf2n &o) { return (*this += o); }
    gf2n operator+(const gf2n &o) const { gf2n r(*this); r += o; return r; }
    gf2n operator-(const gf2n &o) const { gf2n r(*this); r += o; return r; }
    gf2n operator-() const { return *this; }

    // reduce a 2W-word carry-less product in place into the low W words
    static void reduce(uint64_t *c)
    {
        for (int i = 2 * W - 1; i >= W; --i) {
            const uint64_t t = c[i];
            c[i] = 0;
            // t * x^(64 i) = t * x^(64 (i-W)) * TAIL
            uint64_t lo = 0, hi = 0;
            for (int b = 0; b < 64; ++b) {
                if ((TAIL >> b) & 1) {
                    lo ^= t << b;
                    if (b) hi ^= t >> (64 - b);
                }
            }
            c[i - W] ^= lo;
            c[i - W + 1] ^= hi;
        }
        // the last step may have spilled deg(TAIL) bits back into word W: fold them once more
        const uint64_t t = c[W];
        c[W] = 0;
        for (int b = 0; b < 64; ++b) if ((TAIL >> b) & 1) c[0] ^= t << b;
    }

    gf2n &operator*=(const gf2n &o)
    {
        uint64_t c[2 * W];
        for (int i = 0; i < 2 * W; ++i) c[i] = 0;
        for (int i = 0; i < W; ++i) {
            for (int j = 0; j < W; ++j) {
                uint64_t lo, hi;
                clmul64(w[i], o.w[j], lo, hi);
                c[i + j] ^= lo;
                c[i + j + 1] ^= hi;
            }
        }
        reduce(c);
        for (int i = 0; i < W; ++i) w[i] = c[i];
        return *this;
    }
    gf2n operator*(const gf2n &o) const { gf2n r(*this); r *= o; return r; }

    gf2n squared() const { return (*this) * (*this); }

    // a^(2^n - 2) by square-and-multiply (n = 64 W): inverse of a non-zero element
    gf2n inverse() const
    {
        // exponent 2^n - 2 = 111...10 (n-1 ones then a zero)
        gf2n r = *this;             // a^(1)
        for (int i = 0; i < extension_degree - 2; ++i) {
            r = r.squared();
            r *= *this;             // a^(2^(i+2) - 1)
        }
        return r.squared();         // a^(2^n - 2)
    }

    gf2n pow(uint64_t e) const
    {
        gf2n r = one(), b = *this;
        while (e) { if (e & 1) r *= b; b = b.squared(); e >>= 1; }
        return r;
    }
};

typedef gf2n<1, 0x1Bull>  gf64;     // x^4 + x^3 + x + 1
typedef gf2n<2, 0x87ull>  gf128;    // x^7 + x^2 + x + 1
typedef gf2n<3, 0x87ull>  gf192;    // x^7 + x^2 + x + 1
typedef gf2n<4, 0x425ull> gf256;    // x^10 + x^5 + x^2 + 1

static_assert(sizeof(gf64) == 8 && sizeof(gf192) == 24 && sizeof(gf256) == 32, "raw layout = words");

} // namespace oracle
