// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's Fractal preprocessing SNARK (BASELINE config 5), non-zk, BLAKE2b:
//   libiop/algebra/polynomials/{lagrange_polynomial,bivariate_lagrange_polynomial}.tcc
//   libiop/protocols/encoded/r1cs_rs_iop/fractal_indexer.tcc            matrix_indexer (row, col, val, row*col over K)
//   libiop/protocols/encoded/lincheck/{holographic_lincheck,holographic_lincheck_aux,common}.tcc
//   libiop/protocols/encoded/sumcheck/rational_sumcheck.tcc             reextension + constraint oracle
//   libiop/protocols/encoded/common/{rational_linear_combination,boundary_constraint}.tcc
//   libiop/protocols/fractal_hiop.tcc, libiop/snark/fractal_snark.tcc, libiop/bcs/bcs_indexer.tcc
// The indexer / prover / verifier split of the reference (three bcs_protocol subclasses exchanging index objects) is one
// bcs_protocol here: the indexer runs the one-round protocol and keeps the oracles and the tree roots, the prover resubmits
// the oracles as round 0 (the same trees, bcs_prover.tcc:69-80 takes them from the index), the verifier gets the roots.
// Independent of libiop_amd/.  Citations are relative to /root/reference.
#pragma once
#include "aurora.hpp"

namespace oracle {

// vanishing_polynomial::formal_derivative_at_point (vanishing_polynomial.tcc:55-74)
template<typename F> F vp_formal_derivative(const vanishing_polynomial<F, affine_subspace<F>> &Z, const F &) { return Z.lin[1]; }
template<typename F> F vp_formal_derivative(const vanishing_polynomial<F, mult_coset<F>> &Z, const F &x)
{
    return F((uint64_t)Z.degree) * x.pow(Z.degree - 1);
}

// lagrange_polynomial.tcc:4-136: f(y) = (Z_S(x) - Z_S(y)) / (x - y), optionally normalised by 1 / (DZ_S)(x)
template<typename F>
struct lagrange_polynomial {
    typedef domain_of<F> D;
    F x;
    D S;
    vanishing_polynomial<F, D> Z;
    bool normalized;
    F Z_at_x, normalization;
    lagrange_polynomial(const F &x_, const D &S_, bool normalized_)
        : x(x_), S(S_), Z(S_), normalized(normalized_), Z_at_x(Z.evaluation_at_point(x_)),
          normalization(normalized_ ? vp_formal_derivative<F>(Z, x_).inverse() : F::one()) {}
    F evaluation_at_point(const F &y) const                                                        // :38-64
    {
        if (x == y) return normalized ? F::one() : vp_formal_derivative<F>(Z, x);
        return (Z_at_x - Z.evaluation_at_point(y)) * (x - y).inverse() * normalization;
    }
    std::vector<F> evaluations_over(const D &evaldomain) const                                     // :66-136 (batch inversion there)
    {
        std::vector<F> out;
        for (const F &y : dom_elements(evaldomain)) out.push_back(evaluation_at_point(y));
        return out;
    }
};

// ---- matrix_indexer::compute_oracles_over_K (fractal_indexer.tcc:47-121): {row, col, val, row*col} of M' = M^T scaled ----
template<typename F>
std::vector<std::vector<F>> matrix_index_over_K(const std::vector<typename r1cs_system<F>::row> &M, const domain_of<F> &index_domain,
                                                const domain_of<F> &matrix_domain, size_t input_variable_dim)
{
    typedef domain_of<F> D;
    const vanishing_polynomial<F, D> Z_H(matrix_domain);
    const size_t K = dom_size(index_domain);
    std::vector<F> row_evals, col_evals, val_evals, row_times_col_evals;
    size_t num_nonzero = 0;
    for (size_t i = 0; i < M.size(); ++i) {
        const F row_index_elem = dom_element(matrix_domain, i);
        for (auto &term : M[i]) {
            ++num_nonzero;
            row_evals.push_back(row_index_elem);
            const F col_index_elem = dom_element(matrix_domain, dom_reindex_by_subset(matrix_domain, input_variable_dim, term.first));
            col_evals.push_back(col_index_elem);
            row_times_col_evals.push_back(row_index_elem * col_index_elem);
            val_evals.push_back(term.second * vp_formal_derivative<F>(Z_H, col_index_elem).inverse());   // u_H(col, col) = (DZ_H)(col)
        }
    }
    if (num_nonzero > K) throw std::invalid_argument("index domain smaller than the number of non-zero entries");
    const F arbitrary_elem_in_H = dom_element(matrix_domain, 0);
    for (size_t i = num_nonzero; i < K; ++i) { row_evals.push_back(arbitrary_elem_in_H); col_evals.push_back(arbitrary_elem_in_H); val_evals.push_back(F::zero()); }
    row_evals.swap(col_evals);                                                                     // "We are dealing with the transpose"
    const F k0 = dom_element(index_domain, 0);
    row_evals.resize(K, k0);
    col_evals.resize(K, k0);
    val_evals.resize(K, F::zero());
    row_times_col_evals.resize(K, k0 * k0);                                                         // :118-119 pads with K[0]^2
    return { row_evals, col_evals, val_evals, row_times_col_evals };
}

// compute_p_alpha_M (lincheck/common.tcc:5-38)
template<typename F>
std::vector<F> compute_p_alpha_M(size_t input_variable_dim, const domain_of<F> &summation_domain, const std::vector<F> &p_alpha_over_H,
                                 const std::vector<F> &r_Mz, const std::vector<const std::vector<typename r1cs_system<F>::row> *> &matrices)
{
    std::vector<F> over_H(dom_size(summation_domain), F::zero());
    for (size_t m = 0; m < matrices.size(); ++m)
        for (size_t i = 0; i < dom_size(summation_domain); ++i)
            for (auto &term : (*matrices[m])[i])
                over_H[dom_reindex_by_subset(summation_domain, input_variable_dim, term.first)] += r_Mz[m] * term.second * p_alpha_over_H[i];
    return IFFT_over<F>(over_H, summation_domain);
}

// ---- virtual oracles ----
// holographic_lincheck_aux.tcc:4-95: p(alpha, x) * sum_m r_m f_Mz(x) - f_z(x) * t(x); constituents (fz, Mz..., t)
template<typename F>
struct holographic_multi_lincheck_virtual_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    D codeword_domain, summation_domain;
    size_t num_matrices;
    std::vector<F> r_Mz;
    std::shared_ptr<lagrange_polynomial<F>> p_alpha_prime;
    holographic_multi_lincheck_virtual_oracle(const D &L, const D &H, size_t matrices) : codeword_domain(L), summation_domain(H), num_matrices(matrices) {}
    void set_challenge(const F &alpha, const std::vector<F> &r)
    {
        if (r.size() != num_matrices) throw std::invalid_argument("Not enough random linear combination coefficients were provided");
        r_Mz = r;
        p_alpha_prime = std::make_shared<lagrange_polynomial<F>>(alpha, summation_domain, false);
    }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != num_matrices + 2) throw std::invalid_argument("multi_lincheck uses more constituent oracles than what was provided.");
        const std::vector<F> p = p_alpha_prime->evaluations_over(codeword_domain);
        std::vector<F> out;
        for (size_t i = 0; i < dom_size(codeword_domain); ++i) {
            F combined = F::zero();
            for (size_t m = 0; m < num_matrices; ++m) combined += r_Mz[m] * (*c[m + 1])[i];
            out.push_back(combined * p[i] - (*c[0])[i] * (*c[num_matrices + 1])[i]);
        }
        return out;
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override
    {
        if (c.size() != num_matrices + 2) throw std::invalid_argument("multi_lincheck uses more constituent oracles than what was provided.");
        F combined = F::zero();
        for (size_t m = 0; m < num_matrices; ++m) combined += r_Mz[m] * c[m + 1];
        return combined * p_alpha_prime->evaluation_at_point(x) - c[0] * c[num_matrices + 1];
    }
};

// holographic_lincheck_aux.tcc:97-169: (row(x) - row_query)(col(x) - col_query) from (row, col, row*col)
template<typename F>
struct single_matrix_denominator : virtual_oracle<F> {
    F row_query_point = F::zero(), column_query_point = F::zero();
    void set_challenge(const F &row_q, const F &col_q) { row_query_point = row_q; column_query_point = col_q; }
    F combine(const F &row, const F &col, const F &row_col) const
    {
        return (-column_query_point * row) - (row_query_point * col) + row_col + row_query_point * column_query_point;
    }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("single_matrix_denominator was expecting row, col, row*col oracles as input");
        std::vector<F> out;
        for (size_t i = 0; i < c[0]->size(); ++i) out.push_back(combine((*c[0])[i], (*c[1])[i], (*c[2])[i]));
        return out;
    }
    F evaluation_at_point(size_t, const F &, const std::vector<F> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("single_matrix_denominator was expecting row, col, row*col oracles as input");
        return combine(c[0], c[1], c[2]);
    }
};

// rational_linear_combination.tcc:4-52 / :54-134
template<typename F>
struct combined_denominator : virtual_oracle<F> {
    size_t num_rationals;
    explicit combined_denominator(size_t n) : num_rationals(n) {}
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != num_rationals) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        std::vector<F> out(*c[0]);
        for (size_t i = 1; i < c.size(); ++i) {
            if (c[i]->size() != out.size()) throw std::invalid_argument("Vectors of mismatched size.");
            for (size_t j = 0; j < out.size(); ++j) out[j] *= (*c[i])[j];
        }
        return out;
    }
    F evaluation_at_point(size_t, const F &, const std::vector<F> &c) const override
    {
        if (c.size() != num_rationals) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        F r = c[0];
        for (size_t i = 1; i < c.size(); ++i) r *= c[i];
        return r;
    }
};
template<typename F>
struct combined_numerator : virtual_oracle<F> {
    size_t num_rationals;
    std::vector<F> coefficients;
    explicit combined_numerator(size_t n) : num_rationals(n) {}
    void set_coefficients(const std::vector<F> &r)
    {
        if (r.size() != num_rationals) throw std::invalid_argument("Expected same number of random coefficients as oracles.");
        coefficients = r;
    }
    F at(const std::vector<F> &c) const                                                             // (N_0..N_{n-1}, D_0..D_{n-1})
    {
        F result = F::zero();
        for (size_t i = 0; i < num_rationals; ++i) {
            F cur = coefficients[i] * c[i];
            for (size_t j = 0; j < num_rationals; ++j) if (j != i) cur *= c[num_rationals + j];
            result += cur;
        }
        return result;
    }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 2 * num_rationals) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        std::vector<F> out, point(c.size());
        for (size_t j = 0; j < c[0]->size(); ++j) {
            for (size_t i = 0; i < c.size(); ++i) point[i] = (*c[i])[j];
            out.push_back(at(point));
        }
        return out;
    }
    F evaluation_at_point(size_t, const F &, const std::vector<F> &c) const override
    {
        if (c.size() != 2 * num_rationals) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        return at(c);
    }
};

// rational_linear_combination.tcc:136-212
template<typename F>
struct rational_linear_combination {
    bcs_protocol<F> &IOP;
    size_t num_rationals;
    std::shared_ptr<combined_numerator<F>> numerator;
    std::shared_ptr<combined_denominator<F>> denominator;
    oracle_handle numerator_handle{}, denominator_handle{};
    rational_linear_combination(bcs_protocol<F> &iop, size_t n, size_t domain, const std::vector<oracle_handle> &numerator_handles,
                                const std::vector<oracle_handle> &denominator_handles)
        : IOP(iop), num_rationals(n), numerator(std::make_shared<combined_numerator<F>>(n)), denominator(std::make_shared<combined_denominator<F>>(n))
    {
        if (numerator_handles.size() != n || denominator_handles.size() != n)
            throw std::invalid_argument("Rational Linear Combination: #numerator handles passed in != #denominator handles passed in");
        size_t denominator_degree = 1;
        for (size_t i = 0; i < n; ++i) denominator_degree += IOP.get_oracle_degree(denominator_handles[i]) - 1;
        denominator_handle = IOP.register_virtual_oracle(domain, denominator_degree, denominator_handles, denominator);
        size_t numerator_degree = 0;
        for (size_t i = 0; i < n; ++i)
            numerator_degree = std::max(numerator_degree, IOP.get_oracle_degree(numerator_handles[i]) + denominator_degree - IOP.get_oracle_degree(denominator_handles[i]));
        std::vector<oracle_handle> all(numerator_handles);
        all.insert(all.end(), denominator_handles.begin(), denominator_handles.end());
        numerator_handle = IOP.register_virtual_oracle(domain, numerator_degree, all, numerator);
    }
    void set_coefficients(const std::vector<F> &r) { numerator->set_coefficients(r); }
    std::vector<F> evaluated_contents(const std::vector<std::vector<F>> &numerator_evals, const std::vector<std::vector<F>> &denominator_evals) const   // :183-209
    {
        std::vector<const std::vector<F> *> d, all;
        for (auto &v : denominator_evals) d.push_back(&v);
        for (auto &v : numerator_evals) all.push_back(&v);
        for (auto &v : denominator_evals) all.push_back(&v);
        const std::vector<F> den = denominator->evaluated_contents(d);
        std::vector<F> result = numerator->evaluated_contents(all);
        for (size_t i = 0; i < result.size(); ++i) {
            if (den[i].is_zero()) throw std::invalid_argument("batch_inverse: zero denominator");
            result[i] *= den[i].inverse();
        }
        return result;
    }
};

// boundary_constraint.tcc: (f(x) - claimed_eval) / (x - eval_point)
template<typename F>
struct single_boundary_constraint : virtual_oracle<F> {
    typedef domain_of<F> D;
    D codeword_domain;
    F eval_point = F::zero(), oracle_evaluation = F::zero();
    explicit single_boundary_constraint(const D &L) : codeword_domain(L) {}
    void set_evaluation_point_and_eval(const F &point, const F &eval) { eval_point = point; oracle_evaluation = eval; }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 1) throw std::invalid_argument("Single Boundary Constraint: Expected exactly 1 constituent oracle.");
        const std::vector<F> xs = dom_elements(codeword_domain);
        std::vector<F> out;
        for (size_t i = 0; i < xs.size(); ++i) out.push_back(((*c[0])[i] - oracle_evaluation) * (xs[i] - eval_point).inverse());
        return out;
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override
    {
        if (c.size() != 1) throw std::invalid_argument("Single Boundary Constraint: Expected exactly 1 constituent oracle.");
        return (c[0] - oracle_evaluation) * (x - eval_point).inverse();
    }
};

// rational_sumcheck.tcc:9-137: q = (D (x p + mu / |H|) - N) / Z_H (multiplicative), (D (p + mu / eps x^(|H| - 1)) - N) / Z_H (additive)
template<typename F>
struct sumcheck_constraint_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    D summation_domain, codeword_domain;
    vanishing_polynomial<F, D> Z;
    F claimed_sum = F::zero();
    sumcheck_constraint_oracle(const D &H, const D &L) : summation_domain(H), codeword_domain(L), Z(H) {}
    void set_claimed_sum(const F &mu) { claimed_sum = mu; }
    F shifted_p(const F &x, const F &p, const affine_subspace<F> &H) const { return p + Z.lin[1].inverse() * claimed_sum * x.pow(H.num_elements() - 1); }
    F shifted_p(const F &x, const F &p, const mult_coset<F> &H) const { return p * x + F((uint64_t)H.order).inverse() * claimed_sum; }
    F at(const F &x, const F &p, const F &N, const F &Dn) const { return (Dn * shifted_p(x, p, summation_domain) - N) * Z.evaluation_at_point(x).inverse(); }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("sumcheck_constraint_oracle has three constituent oracles");
        const std::vector<F> xs = dom_elements(codeword_domain);
        std::vector<F> out;
        for (size_t i = 0; i < xs.size(); ++i) out.push_back(at(xs[i], (*c[0])[i], (*c[1])[i], (*c[2])[i]));
        return out;
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("sumcheck_constraint_oracle has three constituent oracles");
        return at(x, c[0], c[1], c[2]);
    }
};

// rational_sumcheck.tcc:139-274
template<typename F>
struct rational_sumcheck_protocol {
    typedef domain_of<F> D;
    bcs_protocol<F> &IOP;
    size_t summation_domain_handle, codeword_domain_handle;
    D summation_domain, codeword_domain;
    size_t reextended_oracle_degree, constraint_oracle_degree;
    oracle_handle numerator_handle{}, denominator_handle{}, reextended_oracle_handle{}, constraint_oracle_handle{};
    std::shared_ptr<sumcheck_constraint_oracle<F>> constraint_oracle;
    F claimed_sum = F::zero();
    rational_sumcheck_protocol(bcs_protocol<F> &iop, size_t summation_h, size_t codeword_h, size_t numerator_degree_bound, size_t denominator_degree_bound)
        : IOP(iop), summation_domain_handle(summation_h), codeword_domain_handle(codeword_h), summation_domain(iop.get_domain(summation_h)),
          codeword_domain(iop.get_domain(codeword_h))
    {
        const size_t n = dom_size(summation_domain);
        reextended_oracle_degree = n - 1;
        constraint_oracle_degree = std::max(numerator_degree_bound, denominator_degree_bound + n - 1) - n;
    }
    void register_summation_oracle(const oracle_handle &numerator, const oracle_handle &denominator) { numerator_handle = numerator; denominator_handle = denominator; }
    void register_proof()
    {
        reextended_oracle_handle = IOP.register_oracle(codeword_domain_handle, reextended_oracle_degree, false);
        constraint_oracle = std::make_shared<sumcheck_constraint_oracle<F>>(summation_domain, codeword_domain);
        constraint_oracle_handle = IOP.register_virtual_oracle(codeword_domain_handle, constraint_oracle_degree,
                                                               { reextended_oracle_handle, numerator_handle, denominator_handle }, constraint_oracle);
    }
    F sum_and_strip(std::vector<F> &coeffs, const affine_subspace<F> &H) const                      // :238-244
    {
        const vanishing_polynomial<F, affine_subspace<F>> Z_H(H);
        const F sum = Z_H.lin[1] * coeffs[H.num_elements() - 1];
        coeffs.pop_back();
        return sum;
    }
    F sum_and_strip(std::vector<F> &coeffs, const mult_coset<F> &H) const                           // :231-236
    {
        const F sum = coeffs[0] * F((uint64_t)H.order);
        coeffs.erase(coeffs.begin());
        return sum;
    }
    void calculate_and_submit_proof(const std::vector<F> &rational_function_over_summation_domain)
    {
        std::vector<F> coeffs = IFFT_over<F>(rational_function_over_summation_domain, summation_domain);
        claimed_sum = sum_and_strip(coeffs, summation_domain);
        IOP.submit_oracle(reextended_oracle_handle, FFT_over<F>(coeffs, codeword_domain));
        constraint_oracle->set_claimed_sum(claimed_sum);
    }
    void construct_verifier_state(const F &sum) { claimed_sum = sum; constraint_oracle->set_claimed_sum(sum); }
    std::vector<oracle_handle> get_all_oracle_handles() const { return { reextended_oracle_handle, constraint_oracle_handle }; }
};

// holographic_lincheck.tcc:4-70 — repetitions
template<typename F>
size_t holographic_lincheck_repetitions(size_t interactive_security_parameter, size_t constraint_domain_dim)
{
    const long double field_bits = (long double)field_info<F>::soundness_log_of_field_size();
    const long double per_repetition = (long double)(1 + constraint_domain_dim) - field_bits;
    return std::max<size_t>(1, (size_t)ceill(-1.0L * (long double)interactive_security_parameter / per_repetition));
}

// ---- holographic multi lincheck (holographic_lincheck.tcc:113-520), non-zk ----
template<typename F>
struct holographic_multi_lincheck {
    typedef domain_of<F> D;
    typedef std::vector<typename r1cs_system<F>::row> matrix;
    bcs_protocol<F> &IOP;
    size_t codeword_domain_handle, summation_domain_handle, index_domain_handle = 0, input_variable_dim, num_matrices, repetitions, lincheck_degree;
    D codeword_domain, summation_domain, index_domain;
    std::vector<const matrix *> matrices;
    std::vector<oracle_handle> constituent_oracle_handles;
    std::vector<std::shared_ptr<batch_sumcheck_protocol<F>>> sumcheck_H;
    std::vector<std::shared_ptr<rational_sumcheck_protocol<F>>> sumcheck_K;
    std::vector<std::shared_ptr<holographic_multi_lincheck_virtual_oracle<F>>> lincheck_oracles;
    std::vector<std::shared_ptr<single_boundary_constraint<F>>> t_boundary_constraint;
    std::vector<std::vector<std::shared_ptr<single_matrix_denominator<F>>>> matrix_denominators;
    std::vector<std::vector<oracle_handle>> matrix_numerator_handles, matrix_denominator_handles;
    std::vector<std::shared_ptr<rational_linear_combination<F>>> rational_lc;
    std::vector<size_t> alpha_handle, random_coefficient_handle, beta_handle, M_at_alpha_beta;
    std::vector<oracle_handle> t_oracle_handle, t_boundary_constraint_handle;
    std::vector<std::vector<F>> r_Mz;

    holographic_multi_lincheck(bcs_protocol<F> &iop, size_t codeword_h, size_t summation_h, size_t input_dim, const std::vector<const matrix *> &M,
                               const oracle_handle &fz_handle, const std::vector<oracle_handle> &Mz_handles, size_t reps)
        : IOP(iop), codeword_domain_handle(codeword_h), summation_domain_handle(summation_h), input_variable_dim(input_dim), num_matrices(M.size()),
          repetitions(reps), codeword_domain(iop.get_domain(codeword_h)), summation_domain(iop.get_domain(summation_h)),
          index_domain(iop.get_domain(summation_h)), matrices(M)
    {
        if (num_matrices < 1) throw std::invalid_argument("multi_lincheck expects at least one matrix");
        if (Mz_handles.size() != num_matrices) throw std::invalid_argument("inconsistent number of Mz_handles and matrices passed into multi lincheck.");
        constituent_oracle_handles.push_back(fz_handle);
        for (auto &h : Mz_handles) constituent_oracle_handles.push_back(h);
        lincheck_degree = dom_size(summation_domain) + std::max(IOP.get_oracle_degree(fz_handle), IOP.get_oracle_degree(Mz_handles[0])) - 1;
        for (size_t r = 0; r < repetitions; ++r) {
            sumcheck_H.push_back(std::make_shared<batch_sumcheck_protocol<F>>(IOP, summation_h, codeword_h, lincheck_degree));
            lincheck_oracles.push_back(std::make_shared<holographic_multi_lincheck_virtual_oracle<F>>(codeword_domain, summation_domain, num_matrices));
            t_boundary_constraint.push_back(std::make_shared<single_boundary_constraint<F>>(codeword_domain));
        }
    }
    void set_index_oracles(size_t indexed_domain_handle, const std::vector<std::vector<oracle_handle>> &indexed_handles)      // :190-254
    {
        if (indexed_handles.size() != num_matrices) throw std::invalid_argument("Incorrect number of sets of indexed oracles");
        for (auto &set : indexed_handles) if (set.size() != 4) throw std::invalid_argument("Incorrect number of indexed oracles within set");
        index_domain_handle = indexed_domain_handle;
        index_domain = IOP.get_domain(indexed_domain_handle);
        const size_t single_degree = dom_size(index_domain);
        const size_t combined_numerator_degree = single_degree + (num_matrices - 1) * single_degree - (num_matrices - 1);
        const size_t combined_denominator_degree = num_matrices * single_degree - (num_matrices - 1);
        matrix_denominators.resize(repetitions);
        matrix_numerator_handles.resize(repetitions);
        matrix_denominator_handles.resize(repetitions);
        for (size_t r = 0; r < repetitions; ++r) {
            for (size_t i = 0; i < num_matrices; ++i) {
                matrix_denominators[r].push_back(std::make_shared<single_matrix_denominator<F>>());
                matrix_numerator_handles[r].push_back(indexed_handles[i][2]);                                              // val
                matrix_denominator_handles[r].push_back(IOP.register_virtual_oracle(codeword_domain_handle, single_degree,
                    { indexed_handles[i][0], indexed_handles[i][1], indexed_handles[i][3] }, matrix_denominators[r][i])); // row, col, row*col
            }
            sumcheck_K.push_back(std::make_shared<rational_sumcheck_protocol<F>>(IOP, index_domain_handle, codeword_domain_handle,
                                                                                 combined_numerator_degree, combined_denominator_degree));
        }
    }
    void register_challenge_alpha()                                                                                         // :256-265
    {
        for (size_t r = 0; r < repetitions; ++r) alpha_handle.push_back(IOP.register_verifier_random_message(1));
        for (size_t r = 0; r < repetitions; ++r) random_coefficient_handle.push_back(IOP.register_verifier_random_message(num_matrices));
    }
    void register_response_alpha()                                                                                          // :267-300
    {
        for (size_t r = 0; r < repetitions; ++r) {
            t_oracle_handle.push_back(IOP.register_oracle(codeword_domain_handle, dom_size(summation_domain), false));
            std::vector<oracle_handle> constituents(constituent_oracle_handles);
            constituents.push_back(t_oracle_handle[r]);
            sumcheck_H[r]->attach_oracle_for_summing(IOP.register_virtual_oracle(codeword_domain_handle, lincheck_degree, constituents, lincheck_oracles[r]));
        }
    }
    void register_challenge_beta()                                                                                          // :302-310
    {
        for (size_t r = 0; r < repetitions; ++r) beta_handle.push_back(IOP.register_verifier_random_message(1));
        for (size_t r = 0; r < repetitions; ++r) sumcheck_H[r]->register_challenge();
    }
    void register_response_beta()                                                                                           // :312-366
    {
        for (size_t r = 0; r < repetitions; ++r) M_at_alpha_beta.push_back(IOP.register_prover_message(1));
        for (size_t r = 0; r < repetitions; ++r) {
            rational_lc.push_back(std::make_shared<rational_linear_combination<F>>(IOP, num_matrices, codeword_domain_handle,
                                                                                   matrix_numerator_handles[r], matrix_denominator_handles[r]));
            sumcheck_K[r]->register_summation_oracle(rational_lc[r]->numerator_handle, rational_lc[r]->denominator_handle);
            t_boundary_constraint_handle.push_back(IOP.register_virtual_oracle(codeword_domain_handle, dom_size(summation_domain) - 1,
                                                                               { t_oracle_handle[r] }, t_boundary_constraint[r]));
            sumcheck_H[r]->register_proof();
            sumcheck_K[r]->register_proof();
        }
    }
    void calculate_response_alpha()                                                                                         // :381-417
    {
        r_Mz.resize(repetitions);
        for (size_t r = 0; r < repetitions; ++r) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handle[r])[0];
            r_Mz[r] = IOP.obtain_verifier_random_message(random_coefficient_handle[r]);
            const lagrange_polynomial<F> p_alpha(alpha, summation_domain, false);
            const std::vector<F> p_alpha_over_H = p_alpha.evaluations_over(summation_domain);
            const std::vector<F> p_alpha_M = compute_p_alpha_M<F>(input_variable_dim, summation_domain, p_alpha_over_H, r_Mz[r], matrices);
            IOP.submit_oracle(t_oracle_handle[r], FFT_over<F>(p_alpha_M, codeword_domain));
            lincheck_oracles[r]->set_challenge(alpha, r_Mz[r]);
        }
    }
    void set_rational_linear_combination_coefficients()                                                                     // :480-499
    {
        const vanishing_polynomial<F, D> Z_H(summation_domain);
        for (size_t r = 0; r < repetitions; ++r) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handle[r])[0], beta = IOP.obtain_verifier_random_message(beta_handle[r])[0];
            const F shift = Z_H.evaluation_at_point(alpha) * Z_H.evaluation_at_point(beta);
            std::vector<F> coefficients;
            for (size_t i = 0; i < num_matrices; ++i) coefficients.push_back(shift * r_Mz[r][i]);
            rational_lc[r]->set_coefficients(coefficients);
        }
    }
    void set_matrix_denominator_challenges()                                                                                // :501-513
    {
        for (size_t r = 0; r < repetitions; ++r) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handle[r])[0], beta = IOP.obtain_verifier_random_message(beta_handle[r])[0];
            for (size_t i = 0; i < num_matrices; ++i) matrix_denominators[r][i]->set_challenge(beta, alpha);
        }
    }
    void calculate_response_beta()                                                                                          // :430-478
    {
        set_rational_linear_combination_coefficients();
        set_matrix_denominator_challenges();
        for (size_t r = 0; r < repetitions; ++r) {
            const F beta = IOP.obtain_verifier_random_message(beta_handle[r])[0];
            std::vector<std::vector<F>> numerators_over_K, denominators_over_K;
            for (size_t i = 0; i < num_matrices; ++i) {
                const std::vector<std::vector<F>> idx = matrix_index_over_K<F>(*matrices[i], index_domain, summation_domain, input_variable_dim);
                numerators_over_K.push_back(idx[2]);
                denominators_over_K.push_back(matrix_denominators[r][i]->evaluated_contents({ &idx[0], &idx[1], &idx[3] }));
            }
            const std::vector<F> combined_rational_over_K = rational_lc[r]->evaluated_contents(numerators_over_K, denominators_over_K);
            sumcheck_K[r]->calculate_and_submit_proof(combined_rational_over_K);
            const F M_value = sumcheck_K[r]->claimed_sum;
            IOP.submit_prover_message(M_at_alpha_beta[r], { M_value });
            t_boundary_constraint[r]->set_evaluation_point_and_eval(beta, M_value);
            sumcheck_H[r]->calculate_and_submit_proof();
        }
    }
    void construct_verifier_state()                                                                                         // :515-548
    {
        r_Mz.resize(repetitions);
        for (size_t r = 0; r < repetitions; ++r) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handle[r])[0];
            r_Mz[r] = IOP.obtain_verifier_random_message(random_coefficient_handle[r]);
            lincheck_oracles[r]->set_challenge(alpha, r_Mz[r]);
            const F beta = IOP.obtain_verifier_random_message(beta_handle[r])[0];
            const F claimed = IOP.receive_prover_message(M_at_alpha_beta[r])[0];
            t_boundary_constraint[r]->set_evaluation_point_and_eval(beta, claimed);
            sumcheck_H[r]->construct_verifier_state();
            sumcheck_K[r]->construct_verifier_state(claimed);
        }
        set_rational_linear_combination_coefficients();
        set_matrix_denominator_challenges();
    }
    std::vector<oracle_handle> get_all_oracle_handles() const                                                               // :550-580
    {
        std::vector<oracle_handle> out;
        for (size_t r = 0; r < repetitions; ++r) {
            out.push_back(t_oracle_handle[r]);
            out.push_back(t_boundary_constraint_handle[r]);
            for (auto &h : sumcheck_H[r]->get_all_oracle_handles()) out.push_back(h);
            for (auto &h : sumcheck_K[r]->get_all_oracle_handles()) out.push_back(h);
        }
        return out;
    }
};

// ---- parameters (fractal_hiop.tcc:5-150, fractal_snark.tcc:63-96), non-zk, heuristic FRI / optimistic-heuristic LDT reducer ----
template<typename F>
struct fractal_parameters {
    size_t security_parameter, RS_extra_dimensions, num_constraints, num_variables, num_inputs;
    size_t matrix_domain_dim, index_domain_dim, codeword_domain_dim;
    size_t pow_bits, query_soundness_error_bits, interactive_soundness_error_bits;
    size_t max_tested_degree_bound, max_LDT_tested_degree_bound, max_constraint_degree_bound, absolute_proximity_parameter;
    size_t holographic_lincheck_repetitions_, num_output_LDT_instances, fri_interactive_repetitions, fri_query_repetitions;
    std::vector<size_t> localization_parameters;

    fractal_parameters(size_t security, size_t RS_extra, size_t localization_parameter, const r1cs_system<F> &cs)
        : security_parameter(security), RS_extra_dimensions(RS_extra), num_constraints(cs.num_constraints()), num_variables(cs.num_variables),
          num_inputs(cs.num_inputs)
    {
        if (num_constraints & (num_constraints - 1)) throw std::invalid_argument("Fractal requires the number of constraints to be a power of two");
        if (num_constraints != num_variables + 1) throw std::invalid_argument("Fractal requires the matrices to be square");
        size_t max_nonzero = 0;
        for (auto *M : { &cs.A, &cs.B, &cs.C }) {
            size_t nnz = 0;
            for (auto &row : *M) nnz += row.size();
            max_nonzero = std::max(max_nonzero, nnz);
        }
        index_domain_dim = ceil_log2(max_nonzero);
        matrix_domain_dim = ceil_log2(num_constraints);
        codeword_domain_dim = ceil_log2(4 * ((size_t)1 << index_domain_dim)) + RS_extra_dimensions;          // fractal_hiop.tcc:38-39
        pow_bits = ceil_log2(num_constraints) + 3;                                                             // fractal_snark.tcc:90-95
        query_soundness_error_bits = security_parameter + 1 - pow_bits;                                       // fractal_hiop.tcc:77-78
        interactive_soundness_error_bits = security_parameter + 3;
        localization_parameters = localization_parameter_to_array(localization_parameter, codeword_domain_dim, RS_extra_dimensions);
        const long double field_bits = (long double)field_info<F>::soundness_log_of_field_size();
        holographic_lincheck_repetitions_ = holographic_lincheck_repetitions<F>(interactive_soundness_error_bits, matrix_domain_dim);
        // r1cs_rs_iop.tcc:56-97 with holographic = true, b = 0: lincheck 3|H| / 4|H| against rowcheck |H| - 1 / 2|H| - 1
        const size_t H = (size_t)1 << matrix_domain_dim;
        max_tested_degree_bound = std::max<size_t>(3 * H, H - 1);
        max_constraint_degree_bound = std::max<size_t>(4 * H, 2 * H - 1);
        size_t total = 0;
        for (size_t l : localization_parameters) total += l;
        const size_t step = (size_t)1 << total;                                                               // next_testable_degree_bound (fri_ldt.tcc:148-163)
        max_LDT_tested_degree_bound = max_tested_degree_bound % step ? max_tested_degree_bound - max_tested_degree_bound % step + step : max_tested_degree_bound;
        const size_t codeword_size = (size_t)1 << codeword_domain_dim;
        if (max_LDT_tested_degree_bound >= codeword_size || max_constraint_degree_bound >= codeword_size) throw std::invalid_argument("degree bounds exceed the codeword domain");
        absolute_proximity_parameter = std::min(codeword_size - max_constraint_degree_bound, codeword_size - max_LDT_tested_degree_bound) - 1;
        num_output_LDT_instances = std::max<size_t>(1, (size_t)ceill(-1.0L * interactive_soundness_error_bits / ((long double)codeword_domain_dim - field_bits)));
        const long double delta = (long double)absolute_proximity_parameter / exp2l((long double)codeword_domain_dim);
        fri_query_repetitions = std::max<size_t>(1, (size_t)ceill(-1.0L * query_soundness_error_bits / log2l(1 - delta)));
        const long double per_interaction = log2l(exp2l((long double)localization_parameters[0]) - 1.0L) - field_bits;
        fri_interactive_repetitions = std::max<size_t>(1, (size_t)ceill(-1.0L * interactive_soundness_error_bits / per_interaction));
    }
};

// ---- fractal_iop (fractal_hiop.tcc:218-346) ----
template<typename F>
struct fractal_iop {
    typedef domain_of<F> D;
    typedef std::vector<typename r1cs_system<F>::row> matrix;
    bcs_protocol<F> &IOP;
    const r1cs_system<F> &cs;
    const fractal_parameters<F> &params;
    size_t index_domain_handle, matrix_domain_handle, codeword_domain_handle, input_variable_dim;
    std::vector<std::vector<oracle_handle>> indexed_handles;                   // per matrix: row, col, val, row*col
    std::shared_ptr<encoded_aurora_protocol<F>> protocol;
    std::shared_ptr<holographic_multi_lincheck<F>> lincheck;
    std::shared_ptr<LDT_instance_reducer<F>> LDT_reducer;

    fractal_iop(bcs_protocol<F> &iop, const r1cs_system<F> &system, const fractal_parameters<F> &p) : IOP(iop), cs(system), params(p)
    {
        const D unshifted = default_domain<F>((size_t)1 << p.codeword_domain_dim);
        index_domain_handle = IOP.register_domain(default_domain<F>((size_t)1 << p.index_domain_dim));
        matrix_domain_handle = IOP.register_domain(default_domain<F>(p.num_constraints));
        codeword_domain_handle = IOP.register_domain(shifted_domain<F>((size_t)1 << p.codeword_domain_dim, dom_element_outside(unshifted)));
        const size_t quotient_map_size = (size_t)1 << p.localization_parameters[0];
        // register_index_oracles (:277-300); libff::log2(num_inputs)
        input_variable_dim = ceil_log2(cs.num_inputs);
        const size_t oracle_degree_bound = (size_t)1 << p.index_domain_dim;
        for (size_t i = 0; i < 3; ++i) {
            std::vector<oracle_handle> hs;
            for (size_t k = 0; k < 4; ++k) hs.push_back(IOP.register_index_oracle(codeword_domain_handle, oracle_degree_bound));
            indexed_handles.push_back(hs);
        }
        IOP.set_round_parameters(quotient_map_size);
        IOP.signal_index_registrations_done();
        // :253-275
        protocol = std::make_shared<encoded_aurora_protocol<F>>(IOP, matrix_domain_handle, matrix_domain_handle, codeword_domain_handle, cs, 0);
        lincheck = std::make_shared<holographic_multi_lincheck<F>>(IOP, codeword_domain_handle, matrix_domain_handle, dom_dim(protocol->input_variable_domain),
            std::vector<const matrix *>{ &cs.A, &cs.B, &cs.C }, protocol->fz_handle,
            std::vector<oracle_handle>{ protocol->fAz_handle, protocol->fBz_handle, protocol->fCz_handle }, p.holographic_lincheck_repetitions_);
        lincheck->set_index_oracles(index_domain_handle, indexed_handles);
        LDT_reducer = std::make_shared<LDT_instance_reducer<F>>(IOP, codeword_domain_handle, p.num_output_LDT_instances, p.max_LDT_tested_degree_bound);
        IOP.set_round_parameters(quotient_map_size);
    }
    void register_interactions()                                                                   // :302-325
    {
        const size_t quotient_map_size = (size_t)1 << params.localization_parameters[0];
        lincheck->register_challenge_alpha();
        IOP.set_round_parameters(quotient_map_size);
        lincheck->register_response_alpha();
        lincheck->register_challenge_beta();
        lincheck->register_response_beta();
        IOP.set_round_parameters(quotient_map_size);
        std::vector<oracle_handle> handles = lincheck->get_all_oracle_handles();                    // r1cs_rs_iop.tcc:650-668
        for (auto &h : { protocol->fw_handle, protocol->fAz_handle, protocol->fBz_handle, protocol->fCz_handle, protocol->rowcheck_handle }) handles.push_back(h);
        LDT_reducer->register_interactions(handles, params.localization_parameters, params.fri_interactive_repetitions, params.fri_query_repetitions);
    }
    void register_queries() { LDT_reducer->register_queries(); }
    // matrix_indexer::compute_oracles x 3 (fractal_indexer.tcc:123-156) in registration order row, col, val, row*col
    std::vector<std::vector<F>> compute_index_oracles() const
    {
        const D index_domain = IOP.get_domain(index_domain_handle), matrix_domain = IOP.get_domain(matrix_domain_handle), codeword_domain = IOP.get_domain(codeword_domain_handle);
        std::vector<std::vector<F>> out;
        for (auto *M : { &cs.A, &cs.B, &cs.C })
            for (auto &over_K : matrix_index_over_K<F>(*M, index_domain, matrix_domain, input_variable_dim))
                out.push_back(FFT_over<F>(IFFT_over<F>(over_K, index_domain), codeword_domain));
        return out;
    }
    void submit_index(std::vector<std::vector<F>> &&oracles)                                        // iop.tcc:309-341 + bcs_{indexer,prover}::signal_index_submissions_done
    {
        if (oracles.size() != 12) throw std::invalid_argument("The IOP prover index provided the wrong number of evaluations");
        for (size_t i = 0; i < 3; ++i)
            for (size_t k = 0; k < 4; ++k) IOP.submit_oracle(indexed_handles[i][k], std::move(oracles[4 * i + k]));
        IOP.signal_prover_round_done();
    }
    void produce_proof(const std::vector<F> &primary_input, const std::vector<F> &auxiliary_input, std::vector<std::vector<F>> &&index_oracles)   // :316-329
    {
        submit_index(std::move(index_oracles));
        protocol->submit_witness_oracles(primary_input, auxiliary_input);
        IOP.signal_prover_round_done();
        lincheck->calculate_response_alpha();                                                       // r1cs_rs_iop.tcc:618-627
        IOP.signal_prover_round_done();
        lincheck->calculate_response_beta();
        IOP.signal_prover_round_done();
        LDT_reducer->calculate_and_submit_proof();
    }
    bool verifier_predicate(const std::vector<F> &primary_input)                                    // :331-343
    {
        protocol->fz_oracle->set_primary_input(primary_input);
        lincheck->construct_verifier_state();
        return LDT_reducer->verifier_predicate();
    }
};

// fractal_snark.tcc:114-133: the index — oracle evaluations for the prover, tree roots for the verifier
template<typename F>
struct fractal_index {
    std::vector<std::vector<F>> oracles;
    std::vector<digest_t> MT_roots;
};
template<typename F>
fractal_index<F> fractal_snark_indexer(const r1cs_system<F> &cs, const fractal_parameters<F> &params)
{
    bcs_protocol<F> IOP(params.pow_bits);
    fractal_iop<F> full_protocol(IOP, cs, params);
    IOP.seal_interaction_registrations();
    IOP.seal_query_registrations();
    fractal_index<F> index;
    index.oracles = full_protocol.compute_index_oracles();
    full_protocol.submit_index(std::vector<std::vector<F>>(index.oracles));
    index.MT_roots = IOP.get_index_MT_roots();
    return index;
}

// fractal_snark.tcc:135-162
template<typename F>
bcs_transcript<F> fractal_snark_prover(const fractal_index<F> &index, const r1cs_system<F> &cs, const std::vector<F> &primary_input,
                                       const std::vector<F> &auxiliary_input, const fractal_parameters<F> &params)
{
    bcs_protocol<F> IOP(params.pow_bits);
    fractal_iop<F> full_protocol(IOP, cs, params);
    full_protocol.register_interactions();
    IOP.seal_interaction_registrations();
    full_protocol.register_queries();
    IOP.seal_query_registrations();
    full_protocol.produce_proof(primary_input, auxiliary_input, std::vector<std::vector<F>>(index.oracles));
    return IOP.get_transcript();
}

// fractal_snark.tcc:164-197
template<typename F>
bool fractal_snark_verifier(const std::vector<digest_t> &index_MT_roots, const r1cs_system<F> &cs, const std::vector<F> &primary_input,
                            const bcs_transcript<F> &proof, const fractal_parameters<F> &params)
{
    try {
        bcs_protocol<F> IOP(params.pow_bits, proof, index_MT_roots);
        fractal_iop<F> full_protocol(IOP, cs, params);
        full_protocol.register_interactions();
        IOP.seal_interaction_registrations();
        full_protocol.register_queries();
        IOP.seal_query_registrations();
        if (!IOP.transcript_is_valid()) return false;
        return full_protocol.verifier_predicate(primary_input);
    } catch (const std::exception &) {
        return false;
    }
}

} // namespace oracle
