// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libfqfft's radix-2 helpers (libfqfft is an empty submodule of the reference tree).
#pragma once
#include <vector>
#include <libff/common/utils.hpp>
#include <libff/algebra/field_utils/field_utils.hpp>

namespace libfqfft {

// in-place decimation-in-time transform: a[i] <- sum_j a[j] omega^(i j)
template<typename FieldT>
void _basic_serial_radix2_FFT(std::vector<FieldT> &a, const FieldT &omega)
{
    const std::size_t n = a.size(), logn = libff::log2(n);
    if (n != ((std::size_t)1 << logn)) throw std::invalid_argument("expected n == (1u << logn)");
    for (std::size_t k = 0; k < n; ++k) {
        const std::size_t rk = libff::bitreverse(k, logn);
        if (k < rk) std::swap(a[k], a[rk]);
    }
    std::size_t m = 1;
    for (std::size_t s = 1; s <= logn; ++s) {
        const FieldT w_m = libff::power(omega, (unsigned long)(n / (2 * m)));
        for (std::size_t k = 0; k < n; k += 2 * m) {
            FieldT w = FieldT::one();
            for (std::size_t j = 0; j < m; ++j) {
                const FieldT t = w * a[k + j + m];
                a[k + j + m] = a[k + j] - t;
                a[k + j] += t;
                w *= w_m;
            }
        }
        m *= 2;
    }
}

template<typename FieldT>
void _multiply_by_coset(std::vector<FieldT> &a, const FieldT &g)
{
    FieldT u = g;
    for (std::size_t i = 1; i < a.size(); ++i) { a[i] *= u; u *= g; }
}

} // namespace libfqfft
