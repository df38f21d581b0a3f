// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libff's bigint<n> — n 64-bit limbs, little endian, the members libiop names.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <gmp.h>

namespace libff {

template<mp_size_t n>
class bigint {
public:
    static const mp_size_t N = n;
    mp_limb_t data[n];
    bigint() { std::memset(data, 0, sizeof(data)); }
    bigint(const unsigned long x) { std::memset(data, 0, sizeof(data)); data[0] = x; }
    bigint(const char *decimal)                              // libff: a decimal string (Poseidon's constants)
    {
        std::memset(data, 0, sizeof(data));
        mpz_t v;
        mpz_init_set_str(v, decimal, 10);
        size_t count = 0;
        mpz_export(data, &count, -1, sizeof(mp_limb_t), 0, 0, v);
        mpz_clear(v);
    }
    bool operator==(const bigint<n> &o) const { return std::memcmp(data, o.data, sizeof(data)) == 0; }
    bool operator!=(const bigint<n> &o) const { return !(*this == o); }
    void clear() { std::memset(data, 0, sizeof(data)); }
    bool is_zero() const { for (mp_size_t i = 0; i < n; ++i) if (data[i]) return false; return true; }
    std::size_t max_bits() const { return n * 64; }
    std::size_t num_bits() const
    {
        for (long i = (long)max_bits() - 1; i >= 0; --i) if (test_bit((std::size_t)i)) return (std::size_t)i + 1;
        return 0;
    }
    unsigned long as_ulong() const { return data[0]; }
    void to_mpz(mpz_t r) const
    {
        mpz_set_ui(r, 0);
        for (long i = n - 1; i >= 0; --i) { mpz_mul_2exp(r, r, 64); mpz_add_ui(r, r, data[i]); }
    }
    bool test_bit(const std::size_t bitno) const { return bitno < (std::size_t)n * 64 && ((data[bitno / 64] >> (bitno % 64)) & 1); }
};

} // namespace libff
