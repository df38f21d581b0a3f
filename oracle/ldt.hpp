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

} // namespace oracle
