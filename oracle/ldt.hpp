// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the LDT reducer's combined virtual oracle and the element-power helper under it:
//   libiop/algebra/exponentiation.tcc:3-91          subset_element_powers
//   libiop/protocols/ldt/ldt_reducer_aux.tcc:3-136  combined_LDT_virtual_oracle
// Outputs are unique field elements; the reference's tests check them through identities only
// (tests/algebra/test_exponentiation.cpp, tests/protocols/test_ldt_reducer.cpp), which tests/ repeat on this code.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>
#include "algebra.hpp"
#include "mult.hpp"

namespace oracle {

// exponentiation.tcc:3-19
template<typename F>
std::vector<F> subspace_to_power_of_two(const affine_subspace<F> &S, uint64_t power_of_two)
{
    std::vector<F> basis_powers(S.basis);
    for (F &el : basis_powers) el = el.pow(power_of_two);
    return all_subset_sums<F>(basis_powers, S.shift.pow(power_of_two));
}

// exponentiation.tcc:21-56
template<typename F>
std::vector<F> subspace_element_powers(const affine_subspace<F> &S, uint64_t exponent)
{
    if (exponent != 0 && (exponent & (exponent - 1)) == 0) return subspace_to_power_of_two(S, exponent);
    std::vector<F> result(S.num_elements(), F::one());
    for (size_t i = 0; i < 64; ++i) {
        if (!(exponent & (1ull << i))) continue;
        const std::vector<F> t = subspace_to_power_of_two(S, 1ull << i);
        for (size_t j = 0; j < result.size(); ++j) result[j] *= t[j];
    }
    return result;
}

// exponentiation.tcc:58-73
template<typename F>
std::vector<F> coset_element_powers(const mult_coset<F> &S, uint64_t exponent)
{
    std::vector<F> result;
    const F g_to_exp = S.g.pow(exponent);
    F cur = S.shift.pow(exponent);
    for (size_t i = 0; i < S.order; ++i) { result.push_back(cur); cur *= g_to_exp; }
    return result;
}

// ldt_reducer_aux.tcc:3-136
template<typename F>
struct combined_LDT_virtual_oracle {
    std::vector<size_t> degrees, submaximal, maximal;
    size_t max_degree;
    std::vector<F> coefficients;

    explicit combined_LDT_virtual_oracle(const std::vector<size_t> &input_oracle_degrees) : degrees(input_oracle_degrees)
    {
        max_degree = *std::max_element(degrees.begin(), degrees.end());                     // :12
        for (size_t i = 0; i < degrees.size(); ++i) (degrees[i] < max_degree ? submaximal : maximal).push_back(i);
    }
    void set_random_coefficients(const std::vector<F> &r)                                   // :26-37
    {
        if (r.size() != 2 * degrees.size()) throw std::invalid_argument("Expected the nunmber of random coefficients to be twice the number of oracles.");
        coefficients = { F::one() };
        coefficients.insert(coefficients.end(), r.begin(), r.end());
    }
    template<typename BumpFn>
    std::vector<F> combine(const std::vector<std::vector<F>> &evals, BumpFn bump) const     // :39-131
    {
        if (evals.size() != degrees.size()) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        std::vector<F> result(evals[0].size(), F::zero());
        for (size_t index : maximal) {
            if (evals[index].size() != result.size()) throw std::invalid_argument("Vectors of mismatched size.");
            for (size_t j = 0; j < result.size(); ++j) result[j] += coefficients[index] * evals[index][j];
        }
        for (size_t i = 0; i < submaximal.size(); ++i) {
            const size_t index = submaximal[i];
            const std::vector<F> b = bump(max_degree - degrees[index], coefficients[degrees.size() + i]);
            for (size_t j = 0; j < result.size(); ++j) result[j] += (coefficients[index] + b[j]) * evals[index][j];
        }
        return result;
    }
    // additive: bump factor r * x^shift from subset_element_powers (:78-103)
    std::vector<F> evaluated_contents(const affine_subspace<F> &domain, const std::vector<std::vector<F>> &evals) const
    {
        return combine(evals, [&](uint64_t e, const F &r) {
            std::vector<F> b = subspace_element_powers(domain, e);
            for (F &v : b) v = r * v;
            return b;
        });
    }
    // multiplicative: running product r * shift^e * (g^e)^j (:104-128)
    std::vector<F> evaluated_contents(const mult_coset<F> &domain, const std::vector<std::vector<F>> &evals) const
    {
        return combine(evals, [&](uint64_t e, const F &r) {
            std::vector<F> b;
            F cur = r * domain.shift.pow(e);
            const F inc = domain.g.pow(e);
            for (size_t j = 0; j < domain.order; ++j) { b.push_back(cur); cur *= inc; }
            return b;
        });
    }
    // :133-170
    F evaluation_at_point(const F &x, const std::vector<F> &vals) const
    {
        F result = F::zero();
        for (size_t i = 0; i < vals.size(); ++i) result += coefficients[i] * vals[i];
        for (size_t i = 0; i < submaximal.size(); ++i)
            result += coefficients[degrees.size() + i] * x.pow(max_degree - degrees[submaximal[i]]) * vals[submaximal[i]];
        return result;
    }
};

// libiop/protocols/encoded/common/rowcheck.tcc:16-88 — rowcheck_ABC_virtual_oracle::evaluated_contents over an affine
// subspace: Z_H^-1 (Az Bz - Cz), Z_H taken at its |L| / |H| unique evaluations (vanishing_polynomial.tcc:77-95), one per
// contiguous coset of H = span(first h basis vectors) + constraint_shift
template<typename F>
std::vector<F> rowcheck_additive(const std::vector<F> &Az, const std::vector<F> &Bz, const std::vector<F> &Cz,
                                 const affine_subspace<F> &codeword_domain, size_t h, const F &constraint_shift)
{
    const affine_subspace<F> H(std::vector<F>(codeword_domain.basis.begin(), codeword_domain.basis.begin() + h), constraint_shift);
    const std::vector<F> Z = vanishing_polynomial_from_subspace<F>(H);
    const size_t order_H = (size_t)1 << h, n = codeword_domain.num_elements(), num_cosets = n / order_H;
    std::vector<F> z_unique;
    for (size_t i = 0; i < num_cosets; ++i) z_unique.push_back(linearized_eval<F>(Z, codeword_domain.element_by_index(i * order_H)));
    const std::vector<F> Z_inv = batch_inverse_and_mul<F>(z_unique, F::one());
    std::vector<F> result;
    for (size_t i = 0; i < num_cosets; ++i)                                             // :67-82
        for (size_t pos = i * order_H; pos < (i + 1) * order_H; ++pos) result.push_back(Z_inv[i] * (Az[pos] * Bz[pos] - Cz[pos]));
    return result;
}

// the multiplicative arm (:50-65): Z_H(x) = x^|H| - shift_H^|H|, position i * num_cosets + j lies in coset j
template<typename F>
std::vector<F> rowcheck_multiplicative(const std::vector<F> &Az, const std::vector<F> &Bz, const std::vector<F> &Cz,
                                       const mult_coset<F> &codeword_domain, size_t order_H, const F &constraint_shift)
{
    const size_t n = codeword_domain.order, num_cosets = n / order_H;
    const F vp_shift = constraint_shift.pow(order_H);
    std::vector<F> z_unique;
    F cur = codeword_domain.shift;
    for (size_t j = 0; j < num_cosets; ++j) { z_unique.push_back(cur.pow(order_H) - vp_shift); cur *= codeword_domain.g; }
    const std::vector<F> Z_inv = batch_inverse_and_mul<F>(z_unique, F::one());
    std::vector<F> result;
    for (size_t i = 0; i < order_H; ++i)
        for (size_t j = 0; j < num_cosets; ++j) { const size_t pos = i * num_cosets + j; result.push_back(Z_inv[j] * (Az[pos] * Bz[pos] - Cz[pos])); }
    return result;
}

// libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.tcc:181-222 — fz_virtual_oracle::evaluated_contents given f_1v already
// evaluated over the codeword domain (the reference gets it with IFFT_over_field_subset / FFT_over_field_subset, :207-212)
template<typename F>
std::vector<F> fz_additive(const std::vector<F> &fw, const std::vector<F> &f1v, const affine_subspace<F> &codeword_domain,
                           const affine_subspace<F> &input_domain)
{
    const std::vector<F> Z = vanishing_polynomial_from_subspace<F>(input_domain);
    const std::vector<F> pts = codeword_domain.all_elements();
    std::vector<F> result;
    for (size_t i = 0; i < pts.size(); ++i) result.push_back(fw[i] * linearized_eval<F>(Z, pts[i]) + f1v[i]);     // :216-220
    return result;
}

template<typename F>
std::vector<F> fz_multiplicative(const std::vector<F> &fw, const std::vector<F> &f1v, const mult_coset<F> &codeword_domain,
                                 size_t input_order, const F &input_shift)
{
    const F vp_shift = input_shift.pow(input_order);
    const std::vector<F> pts = codeword_domain.all_elements();
    std::vector<F> result;
    for (size_t i = 0; i < pts.size(); ++i) result.push_back(fw[i] * (pts[i].pow(input_order) - vp_shift) + f1v[i]);
    return result;
}

// libiop/protocols/encoded/sumcheck/sumcheck.tcc:58-119 with sumcheck_aux.tcc:3-32 — sumcheck_g_oracle::evaluated_contents
template<typename F>
std::vector<F> sumcheck_g_additive(const std::vector<F> &f, const std::vector<F> &h, const affine_subspace<F> &codeword_domain,
                                   const affine_subspace<F> &summation_domain, const F &claimed_sum)
{
    const std::vector<F> Z = vanishing_polynomial_from_subspace<F>(summation_domain);
    const F eps_inv_times_claimed_sum = Z[1].inverse() * claimed_sum;                      // :36-38, :52-54
    const size_t order_H = summation_domain.num_elements();
    // constant_times_subspace_to_order_H_minus_1 (sumcheck_aux.tcc:3-32)
    const std::vector<F> x_to_H = subspace_element_powers<F>(codeword_domain, order_H);
    std::vector<F> elems = codeword_domain.all_elements();
    std::vector<size_t> zeros;                                                              // utils.tcc:79-97
    for (size_t i = 0; i < elems.size(); ++i) if (elems[i] == F::zero()) { zeros.push_back(i); elems[i] = F::one(); }
    std::vector<F> x_inv_c = batch_inverse_and_mul<F>(elems, eps_inv_times_claimed_sum);
    for (size_t i : zeros) x_inv_c[i] = F::zero();
    const std::vector<F> pts = codeword_domain.all_elements();
    std::vector<F> result(f);
    for (size_t i = 0; i < result.size(); ++i) result[i] -= (x_to_H[i] * x_inv_c[i] + linearized_eval<F>(Z, pts[i]) * h[i]);   // :88-92
    return result;
}

template<typename F>
std::vector<F> sumcheck_g_multiplicative(const std::vector<F> &f, const std::vector<F> &h, const mult_coset<F> &codeword_domain,
                                         size_t order_H, const F &summation_shift, const F &claimed_sum)
{
    const F c = F((uint64_t)order_H).inverse() * claimed_sum;                               // :46-49
    const F vp_shift = summation_shift.pow(order_H);
    F cur_x_inv = codeword_domain.shift.inverse();
    const F g_inv = codeword_domain.g.inverse();
    const std::vector<F> pts = codeword_domain.all_elements();
    std::vector<F> result(f);
    for (size_t i = 0; i < result.size(); ++i) {                                            // :107-115
        result[i] -= (c + (pts[i].pow(order_H) - vp_shift) * h[i]);
        result[i] *= cur_x_inv;
        cur_x_inv *= g_inv;
    }
    return result;
}

// libiop/protocols/encoded/lincheck/basic_lincheck_aux.tcc:102-144 — multi_lincheck_virtual_oracle::evaluated_contents given
// p_alpha^1 / p_alpha^2 over the codeword domain (the reference extends them with FFT_over_field_subset, :112-118)
template<typename F>
std::vector<F> lincheck_combine(const std::vector<F> &fz, const std::vector<std::vector<F>> &Mz, const std::vector<F> &r_Mz,
                                const std::vector<F> &p_alpha_prime, const std::vector<F> &p_alpha_ABC)
{
    const size_t n = fz.size();
    std::vector<F> f_combined_Mz(n, F::zero());
    for (size_t i = 0; i < n; ++i) for (size_t m = 0; m < Mz.size(); ++m) f_combined_Mz[i] += r_Mz[m] * Mz[m][i];       // :124-128
    std::vector<F> result;
    for (size_t i = 0; i < n; ++i) result.push_back(f_combined_Mz[i] * p_alpha_prime[i] - fz[i] * p_alpha_ABC[i]);      // :136-141
    return result;
}

} // namespace oracle
