// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's FRI prover fold and domain chain for additive (binary-field)
// domains.  Citations are relative to /root/reference.
#pragma once
#include "algebra.hpp"

namespace oracle {

// libiop/protocols/ldt/fri/fri_ldt.tcc:132-146
static inline std::vector<size_t> localization_parameter_to_array(size_t localization_parameter,
                                                                  size_t codeword_domain_dim,
                                                                  size_t RS_extra_dimensions)
{
    const size_t num_reductions = ((codeword_domain_dim - RS_extra_dimensions - 1) / localization_parameter) + 1;
    std::vector<size_t> out(num_reductions - 1, localization_parameter);
    out.insert(out.begin(), 1);
    return out;
}

// libiop/protocols/ldt/fri/fri_aux.tcc:36-103 — per coset j (contiguous coset_size elements):
// interpolate f_i on the coset and evaluate at x_i, one batch inversion per coset.
template<typename F>
std::vector<F> additive_evaluate_next_f_i_over_entire_domain(const std::vector<F> &f_i_evals,
                                                             const affine_subspace<F> &f_i_domain,
                                                             size_t coset_size, const F &x_i)
{
    const std::vector<F> all_elements = f_i_domain.all_elements();                    // :43
    const size_t num_cosets = all_elements.size() / coset_size;
    std::vector<F> next;
    next.reserve(num_cosets);

    const affine_subspace<F> unshifted_coset(f_i_domain.subset_of_order(coset_size).basis, F::zero()); // :57-58
    const std::vector<F> unshifted_vp = vanishing_polynomial_from_subspace<F>(unshifted_coset);      // :59-60
    const F unshifted_vp_x = linearized_eval<F>(unshifted_vp, x_i);                                   // :62
    const F inv_vp_linear_term = unshifted_vp[1].inverse();                                           // :63

    std::vector<F> shifted(coset_size);
    for (size_t j = 0; j < num_cosets; ++j) {
        const F coset_shift = all_elements[coset_size * j];                                            // :72
        const F shifted_vp_x = unshifted_vp_x - linearized_eval<F>(unshifted_vp, coset_shift);        // :73-74
        const bool x_in_domain = (shifted_vp_x == F::zero());
        F interpolation = F::zero();
        for (size_t k = 0; k < coset_size; ++k) {
            if (x_in_domain && x_i == all_elements[j * coset_size + k]) {                              // :80-84
                interpolation = f_i_evals[j * coset_size + k];
                break;
            }
            shifted[k] = x_i - all_elements[j * coset_size + k];                                       // :86
        }
        if (!x_in_domain) {
            const F k = inv_vp_linear_term * shifted_vp_x;                                             // :90
            const std::vector<F> lagrange = batch_inverse_and_mul<F>(shifted, k);                     // :91-92
            for (size_t kk = 0; kk < coset_size; ++kk) interpolation += f_i_evals[j * coset_size + kk] * lagrange[kk];
        }
        next.push_back(interpolation);
    }
    return next;
}

// libiop/protocols/ldt/fri/fri_aux.tcc:270-303 — additive_evaluate_next_f_i_at_coset: the verifier's single-coset fold.
//   coset_basis : the first eta basis vectors of L^(i) (the unshifted coset / localizer domain)
//   shift       : the queried coset's first element
template<typename F>
F additive_evaluate_next_f_i_at_coset(const std::vector<F> &f_i_evals_over_coset, const std::vector<F> &coset_basis,
                                      const F &shift, const F &x_i)
{
    const std::vector<F> unshifted_vp = vanishing_polynomial_from_subspace<F>(affine_subspace<F>(coset_basis, F::zero()));
    const F vp_x = linearized_eval<F>(unshifted_vp, x_i) - linearized_eval<F>(unshifted_vp, shift);   // :281-282
    const F c = unshifted_vp[1].inverse();                                                            // :283
    const bool x_in_domain = (vp_x == F::zero());
    const std::vector<F> coset_elems = all_subset_sums<F>(coset_basis, x_i + shift);                  // :286-287
    if (x_in_domain) {
        for (size_t k = 0; k < f_i_evals_over_coset.size(); ++k)
            if (coset_elems[k] == F::zero()) return f_i_evals_over_coset[k];                          // :288-296
    }
    const std::vector<F> lagrange = batch_inverse_and_mul<F>(coset_elems, vp_x * c);                 // :297
    F interpolation = F::zero();
    for (size_t k = 0; k < coset_elems.size(); ++k) interpolation += lagrange[k] * f_i_evals_over_coset[k];
    return interpolation;
}

// libiop/protocols/ldt/fri/fri_ldt.tcc:310-338 — additive domain chain: L^(i+1) has
// basis q(basis[eta..]) and shift q(shift), q = vanishing polynomial of span(basis[0..eta)).
template<typename F>
std::vector<affine_subspace<F>> fri_additive_domains(const affine_subspace<F> &codeword_domain,
                                                     const std::vector<size_t> &localization_parameters)
{
    std::vector<affine_subspace<F>> domains;
    domains.push_back(codeword_domain);
    for (size_t i = 0; i < localization_parameters.size(); ++i) {
        const size_t eta = localization_parameters[i];
        const affine_subspace<F> &last = domains[i];
        const affine_subspace<F> localizer(std::vector<F>(last.basis.begin(), last.basis.begin() + eta), F::zero());
        const std::vector<F> q = vanishing_polynomial_from_subspace<F>(localizer);
        const F next_shift = linearized_eval<F>(q, last.shift);
        std::vector<F> next_basis(last.basis.begin() + eta, last.basis.end());
        for (F &el : next_basis) el = linearized_eval<F>(q, el);
        domains.push_back(affine_subspace<F>(next_basis, next_shift));
    }
    return domains;
}

// Index maps (libiop/algebra/field_subset/subspace.tcc:73-91 additive; subgroup.tcc:175-197
// multiplicative), as used by calculate_next_coset_query_positions (fri_aux.tcc:355-387).
static inline size_t coset_index(bool additive, size_t n, size_t position, size_t coset_size)
{
    return additive ? position / coset_size : position % (n / coset_size);
}
static inline size_t intra_coset_index(bool additive, size_t n, size_t position, size_t coset_size)
{
    return additive ? position % coset_size : position / (n / coset_size);
}
static inline size_t position_by_coset_indices(bool additive, size_t n, size_t cidx, size_t intra, size_t coset_size)
{
    return additive ? cidx * coset_size + intra : cidx + intra * (n / coset_size);
}

// fri_aux.tcc:355-387 with the lambda evaluated directly
static inline std::vector<size_t> next_coset_query_positions(bool additive, size_t non_localized_n, size_t localized_n,
                                                             size_t seed_position, size_t prev_loc, size_t cur_loc)
{
    const size_t prev_cs = (size_t)1 << prev_loc, cur_cs = (size_t)1 << cur_loc;
    std::vector<size_t> out(cur_cs);
    for (size_t i = 0; i < cur_cs; ++i) {
        const size_t localized_position = coset_index(additive, non_localized_n, seed_position, prev_cs);
        const size_t localized_coset = coset_index(additive, localized_n, localized_position, cur_cs);
        out[i] = position_by_coset_indices(additive, localized_n, localized_coset, i, cur_cs);
    }
    return out;
}

} // namespace oracle
