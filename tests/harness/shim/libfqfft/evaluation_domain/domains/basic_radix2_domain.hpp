// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libfqfft::basic_radix2_domain (the members libiop names).
#pragma once
#include <libfqfft/evaluation_domain/domains/basic_radix2_domain_aux.hpp>

namespace libfqfft {

template<typename FieldT>
class basic_radix2_domain {
public:
    std::size_t m;
    FieldT omega;
    basic_radix2_domain(const std::size_t m_) : m(m_)
    {
        if (m <= 1 || (m & (m - 1))) throw std::invalid_argument("basic_radix2(): expected m > 1 and a power of two");
        omega = FieldT::get_root_of_unity(m);      // libff::get_root_of_unity<FieldT>(m)
    }
    void FFT(std::vector<FieldT> &a) const { check(a); _basic_serial_radix2_FFT(a, omega); }
    void iFFT(std::vector<FieldT> &a) const
    {
        check(a);
        _basic_serial_radix2_FFT(a, omega.inverse());
        const FieldT sconst = FieldT(a.size()).inverse();
        for (std::size_t i = 0; i < a.size(); ++i) a[i] *= sconst;
    }
    void cosetFFT(std::vector<FieldT> &a, const FieldT &g) const { _multiply_by_coset(a, g); FFT(a); }
    void icosetFFT(std::vector<FieldT> &a, const FieldT &g) const { iFFT(a); _multiply_by_coset(a, g.inverse()); }
private:
    void check(const std::vector<FieldT> &a) const { if (a.size() != m) throw std::invalid_argument("basic_radix2: expected a.size() == this->m"); }
};

} // namespace libfqfft
