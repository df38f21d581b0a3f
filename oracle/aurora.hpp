// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's Aurora SNARK, prover AND verifier, non-zk, BLAKE2b (configs 1 and 4 of BASELINE.json):
//   libiop/relations/examples/r1cs_examples.tcc:23-78     synthetic R1CS instance (seeded here instead of libsodium)
//   libiop/relations/r1cs.tcc:236-268                     create_Az_Bz_Cz_from_variable_assignment
//   libiop/protocols/aurora_iop.tcc                       parameters, composition, round parameters
//   libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.tcc  witness oracles, fz virtual oracle
//   libiop/protocols/encoded/lincheck/basic_lincheck*.tcc multi_lincheck + its virtual oracle
//   libiop/protocols/encoded/sumcheck/sumcheck*.tcc       batch sumcheck, g oracle
//   libiop/protocols/encoded/common/{rowcheck,random_linear_combination}.tcc
//   libiop/protocols/ldt/ldt_reducer*.tcc                 LDT instance reducer
//   libiop/protocols/ldt/fri/fri_ldt.tcc                  FRI protocol (registration, prover, verifier predicate)
//   libiop/snark/aurora_snark.tcc:119-188                 prover / verifier entry points
// Zero knowledge (make_zk) is out of scope: its masks come from libsodium randomness and are not reproducible.
// Independent of libiop_amd/.  Citations are relative to /root/reference.
#pragma once
#include <cmath>
#include <memory>
#include "iop.hpp"
#include "ldt.hpp"

namespace oracle {

// ---- seeded field elements (SplitMix64 filling canonical elements) ----
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template<int W, uint64_t T> gf2n<W, T> seeded_element(uint64_t seed, uint64_t i, const gf2n<W, T> *)
{
    gf2n<W, T> r;
    for (int k = 0; k < W; ++k) r.w[k] = splitmix64_at(seed, W * i + k);
    return r;
}
template<typename P> Fp<P> seeded_element(uint64_t seed, uint64_t i, const Fp<P> *)       // the 64 N-bit draw reduced mod p
{
    uint64_t c[P::limbs];
    for (int k = 0; k < P::limbs; ++k) c[k] = splitmix64_at(seed, P::limbs * i + k);
    return Fp<P>::from_canonical(c);        // the Montgomery product with R^2 reduces any value below 2^(64 N)
}

// ---- R1CS (relations/r1cs.hpp, variable.hpp): rows of (index, coefficient), index 0 = the constant 1 ----
template<typename F>
struct r1cs_system {
    typedef std::vector<std::pair<size_t, F>> row;
    size_t num_inputs = 0, num_variables = 0;
    std::vector<row> A, B, C;
    size_t num_constraints() const { return A.size(); }
};
template<typename F>
struct r1cs_example {
    r1cs_system<F> cs;
    std::vector<F> primary_input, auxiliary_input;
};

// r1cs_examples.tcc:23-78
template<typename F>
r1cs_example<F> generate_r1cs_example(size_t num_constraints, size_t num_inputs, size_t num_variables, uint64_t seed)
{
    if (num_inputs > num_variables) throw std::invalid_argument("Number of inputs can't exceed number of variables.");
    r1cs_example<F> ex;
    ex.cs.num_inputs = num_inputs;
    ex.cs.num_variables = num_variables;
    std::vector<F> full(num_variables);
    for (size_t i = 0; i < num_variables; ++i) full[i] = seeded_element(seed, i, (const F *)nullptr);
    for (size_t i = 0; i < num_constraints; ++i) {
        const size_t A_idx = i % num_variables, B_idx = (i + 7) % num_variables, C_idx = (2 * i + 1) % num_variables;
        const F AB_val = full[A_idx] * full[B_idx];
        const F C_val = full[C_idx];
        ex.cs.A.push_back({ { A_idx + 1, F::one() } });
        ex.cs.B.push_back({ { B_idx + 1, F::one() } });
        if (C_val.is_zero()) ex.cs.C.push_back({ { 0, AB_val } });
        else ex.cs.C.push_back({ { C_idx + 1, AB_val * C_val.inverse() } });
    }
    ex.primary_input.assign(full.begin(), full.begin() + num_inputs);
    ex.auxiliary_input.assign(full.begin() + num_inputs, full.end());
    return ex;
}

// r1cs.tcc:236-268 — z = (1, primary, auxiliary)
template<typename F>
std::vector<F> sparse_times_vector(const std::vector<typename r1cs_system<F>::row> &M, const std::vector<F> &z)
{
    std::vector<F> out;
    for (auto &row : M) {
        F acc = F::zero();
        for (auto &t : row) acc += z[t.first] * t.second;
        out.push_back(acc);
    }
    return out;
}

// ---- parameters (aurora_iop.tcc:3-186, non-zk; aurora_snark.tcc:38-101; common_bcs_parameters.tcc:9-27) ----
template<typename F>
struct aurora_parameters {
    size_t security_parameter, RS_extra_dimensions, num_constraints, num_variables, num_inputs;
    size_t constraint_domain_dim, variable_domain_dim, summation_domain_dim, codeword_domain_dim;
    size_t pow_bits, query_soundness_error_bits, interactive_soundness_error_bits;
    size_t max_tested_degree_bound, max_constraint_degree_bound, absolute_proximity_parameter;
    size_t multi_lincheck_repetitions, num_output_LDT_instances, fri_interactive_repetitions, fri_query_repetitions;
    std::vector<size_t> localization_parameters;

    aurora_parameters(size_t security, size_t RS_extra, size_t localization_parameter, size_t n_constraints, size_t n_variables, size_t n_inputs)
        : security_parameter(security), RS_extra_dimensions(RS_extra), num_constraints(n_constraints), num_variables(n_variables), num_inputs(n_inputs)
    {
        if (n_constraints & (n_constraints - 1)) throw std::invalid_argument("number of constraints in the constraint system must a power of two.");
        if ((n_variables + 1) & n_variables) throw std::invalid_argument("number of variables in the constraint system must be one less than a power of two.");
        if ((n_inputs + 1) & n_inputs) throw std::invalid_argument("number of inputs in the constraint system must be one less than a power of two.");
        constraint_domain_dim = ceil_log2(n_constraints);
        variable_domain_dim = ceil_log2(n_variables + 1);
        summation_domain_dim = std::max(constraint_domain_dim, variable_domain_dim);
        codeword_domain_dim = summation_domain_dim + RS_extra_dimensions;                          // :37-43, make_zk = false
        pow_bits = constraint_domain_dim + 3;                                                      // default_bcs_params: dim_h + 3 + log2(1)
        query_soundness_error_bits = security_parameter + 1 - pow_bits;                            // :77
        interactive_soundness_error_bits = security_parameter + 3;                                 // :78
        localization_parameters = localization_parameter_to_array(localization_parameter, codeword_domain_dim, RS_extra_dimensions);
        max_tested_degree_bound = (size_t)1 << summation_domain_dim;                               // r1cs_rs_iop.tcc:56-63
        max_constraint_degree_bound = std::max(2 * ((size_t)1 << summation_domain_dim) - 1, 2 * ((size_t)1 << constraint_domain_dim) - 1);   // :82-100
        const long double field_bits = (long double)field_info<F>::soundness_log_of_field_size();
        // basic_lincheck.tcc:52-56
        multi_lincheck_repetitions = std::max<size_t>(1, (size_t)ceill(-1.0L * interactive_soundness_error_bits / ((long double)constraint_domain_dim - field_bits)));
        // ldt_reducer.tcc:19-56, optimistic heuristic
        const size_t codeword_size = (size_t)1 << codeword_domain_dim;
        absolute_proximity_parameter = std::min(codeword_size - max_constraint_degree_bound, codeword_size - max_tested_degree_bound) - 1;
        num_output_LDT_instances = std::max<size_t>(1, (size_t)ceill(-1.0L * interactive_soundness_error_bits / ((long double)codeword_domain_dim - field_bits)));
        // fri_ldt.tcc:8-106, heuristic soundness
        size_t total = 0;
        for (size_t l : localization_parameters) total += l;
        if (max_tested_degree_bound % ((size_t)1 << total)) throw std::invalid_argument("FRI only supports testing degree bounds that are a multiple of 2^{sum of localization parameters}.");
        const long double delta = (long double)absolute_proximity_parameter / exp2l((long double)codeword_domain_dim);
        fri_query_repetitions = std::max<size_t>(1, (size_t)ceill(-1.0L * query_soundness_error_bits / log2l(1 - delta)));
        const long double per_interaction = log2l(exp2l((long double)localization_parameters[0]) - 1.0L) - field_bits;
        fri_interactive_repetitions = std::max<size_t>(1, (size_t)ceill(-1.0L * interactive_soundness_error_bits / per_interaction));
    }
};

// ---- virtual oracles ----
// adapters onto ldt.hpp's two arms
template<typename F> std::vector<F> rowcheck_contents(const std::vector<F> &a, const std::vector<F> &b, const std::vector<F> &c,
                                                      const affine_subspace<F> &L, const affine_subspace<F> &H)
{
    return rowcheck_additive<F>(a, b, c, L, H.dimension(), H.shift);
}
template<typename F> std::vector<F> rowcheck_contents(const std::vector<F> &a, const std::vector<F> &b, const std::vector<F> &c,
                                                      const mult_coset<F> &L, const mult_coset<F> &H)
{
    return rowcheck_multiplicative<F>(a, b, c, L, H.order, H.shift);
}
template<typename F> std::vector<F> fz_contents(const std::vector<F> &fw, const std::vector<F> &f1v, const affine_subspace<F> &L, const affine_subspace<F> &I)
{
    return fz_additive<F>(fw, f1v, L, I);
}
template<typename F> std::vector<F> fz_contents(const std::vector<F> &fw, const std::vector<F> &f1v, const mult_coset<F> &L, const mult_coset<F> &I)
{
    return fz_multiplicative<F>(fw, f1v, L, I.order, I.shift);
}
template<typename F> std::vector<F> sumcheck_g_contents(const std::vector<F> &f, const std::vector<F> &h, const affine_subspace<F> &L,
                                                        const affine_subspace<F> &H, const F &mu)
{
    return sumcheck_g_additive<F>(f, h, L, H, mu);
}
template<typename F> std::vector<F> sumcheck_g_contents(const std::vector<F> &f, const std::vector<F> &h, const mult_coset<F> &L,
                                                        const mult_coset<F> &H, const F &mu)
{
    return sumcheck_g_multiplicative<F>(f, h, L, H.order, H.shift, mu);
}

// r1cs_rs_iop.tcc:141-250
template<typename F>
struct fz_virtual_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    size_t primary_input_size;
    D input_variable_domain, codeword_domain;
    std::vector<F> primary_input;
    bool have_input = false;
    fz_virtual_oracle(size_t k, const D &I, const D &L) : primary_input_size(k), input_variable_domain(I), codeword_domain(L)
    {
        if (dom_size(I) > dom_size(L)) throw std::invalid_argument("Codeword domain must be bigger than the input variable domain.");
    }
    void set_primary_input(const std::vector<F> &p)
    {
        if (p.size() != primary_input_size) throw std::invalid_argument("Primary input size does not match the previously declared size.");
        primary_input = p;
        have_input = true;
    }
    std::vector<F> f_1v_coefficients() const                                                      // :207-210
    {
        std::vector<F> evals = { F::one() };
        evals.insert(evals.end(), primary_input.begin(), primary_input.end());
        return IFFT_over<F>(evals, input_variable_domain);
    }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 1) throw std::invalid_argument("fz_virtual_oracle has one constituent oracle.");
        if (!have_input) throw std::logic_error("Evaluation requested before primary_input is set.");
        if (c[0]->size() != dom_size(codeword_domain)) throw std::invalid_argument("Provided fw evaluations don't match the declared codeword domain size.");
        const std::vector<F> f_1v_over_codeword_domain = FFT_over<F>(f_1v_coefficients(), codeword_domain);      // :211-212
        return fz_contents<F>(*c[0], f_1v_over_codeword_domain, codeword_domain, input_variable_domain);
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override             // :224-249
    {
        if (c.size() != 1) throw std::invalid_argument("fz_virtual_oracle has one constituent oracle.");
        if (!have_input) throw std::logic_error("Evaluation requested before primary_input is set.");
        // f_1v(x): the reference sums Lagrange coefficients of the input domain; the interpolant's value is the same
        const F f1v_X = poly_eval<F>(f_1v_coefficients(), x);
        const vanishing_polynomial<F, D> input_vp(input_variable_domain);
        return c[0] * input_vp.evaluation_at_point(x) + f1v_X;
    }
};

// rowcheck.tcc:5-115
template<typename F>
struct rowcheck_ABC_virtual_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    D codeword_domain, constraint_domain;
    rowcheck_ABC_virtual_oracle(const D &L, const D &H) : codeword_domain(L), constraint_domain(H) {}
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("rowcheck_ABC has three constituent oracles.");
        return rowcheck_contents<F>(*c[0], *c[1], *c[2], codeword_domain, constraint_domain);
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("rowcheck_ABC has three constituent oracles.");
        const vanishing_polynomial<F, D> Z(constraint_domain);
        return (c[0] * c[1] - c[2]) * Z.evaluation_at_point(x).inverse();
    }
};

// basic_lincheck_aux.tcc:5-187
template<typename F>
struct multi_lincheck_virtual_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    typedef std::vector<typename r1cs_system<F>::row> matrix;
    D codeword_domain, constraint_domain, variable_domain, summation_domain;
    size_t input_variable_dim;
    std::vector<const matrix *> matrices;
    std::vector<F> r_Mz, p_alpha_ABC, p_alpha_prime;
    multi_lincheck_virtual_oracle(const D &L, const D &C, const D &V, const D &S, size_t input_dim, const std::vector<const matrix *> &M)
        : codeword_domain(L), constraint_domain(C), variable_domain(V), summation_domain(S), input_variable_dim(input_dim), matrices(M) {}
    void set_challenge(const F &alpha, const std::vector<F> &r)                                    // :29-99
    {
        if (r.size() != matrices.size()) throw std::invalid_argument("Not enough random linear combination coefficients were provided");
        r_Mz = r;
        std::vector<F> alpha_powers;
        F cur = F::one();
        for (size_t i = 0; i < dom_size(constraint_domain); ++i) { alpha_powers.push_back(cur); cur *= alpha; }
        std::vector<F> prime_evals(dom_size(summation_domain), F::zero());
        for (size_t i = 0; i < dom_size(constraint_domain); ++i)
            prime_evals[dom_reindex_by_subset(summation_domain, dom_dim(constraint_domain), i)] = alpha_powers[i];
        std::vector<F> ABC_evals(dom_size(summation_domain), F::zero());
        for (size_t m = 0; m < matrices.size(); ++m)
            for (size_t i = 0; i < dom_size(constraint_domain); ++i)
                for (auto &term : (*matrices[m])[i]) {
                    const size_t variable_index = dom_reindex_by_subset(variable_domain, input_variable_dim, term.first);
                    const size_t summation_index = dom_reindex_by_subset(summation_domain, dom_dim(variable_domain), variable_index);
                    ABC_evals[summation_index] += r_Mz[m] * term.second * alpha_powers[i];
                }
        p_alpha_ABC = IFFT_over<F>(ABC_evals, summation_domain);
        p_alpha_prime = IFFT_over<F>(prime_evals, summation_domain);
    }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override  // :102-144
    {
        if (c.size() != matrices.size() + 1) throw std::invalid_argument("multi_lincheck uses more constituent oracles than what was provided.");
        const std::vector<F> p1 = FFT_over<F>(p_alpha_prime, codeword_domain), p2 = FFT_over<F>(p_alpha_ABC, codeword_domain);
        std::vector<std::vector<F>> Mz;
        for (size_t m = 0; m < matrices.size(); ++m) Mz.push_back(*c[m + 1]);
        return lincheck_combine<F>(*c[0], Mz, r_Mz, p1, p2);
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override              // :146-185 (use_lagrange_ = false)
    {
        if (c.size() != matrices.size() + 1) throw std::invalid_argument("multi_lincheck uses more constituent oracles than what was provided.");
        F combined = F::zero();
        for (size_t i = 0; i < r_Mz.size(); ++i) combined += r_Mz[i] * c[i + 1];
        return combined * poly_eval<F>(p_alpha_prime, x) - c[0] * poly_eval<F>(p_alpha_ABC, x);
    }
};

// random_linear_combination.tcc
template<typename F>
struct random_linear_combination_oracle : virtual_oracle<F> {
    size_t num_oracles;
    std::vector<F> coefficients;
    explicit random_linear_combination_oracle(size_t n) : num_oracles(n) {}
    void set_random_coefficients(const std::vector<F> &r)
    {
        if (r.size() != num_oracles) throw std::invalid_argument("Random Linear Combination Oracle: Expected same number of random coefficients as oracles.");
        coefficients = r;
    }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != num_oracles) throw std::invalid_argument("Random Linear Combination Oracle: Expected same number of evaluations as in registration.");
        std::vector<F> result;
        for (const F &v : *c[0]) result.push_back(coefficients[0] * v);
        for (size_t i = 1; i < c.size(); ++i) {
            if (c[i]->size() != result.size()) throw std::invalid_argument("Vectors of mismatched size.");
            for (size_t j = 0; j < result.size(); ++j) result[j] += coefficients[i] * (*c[i])[j];
        }
        return result;
    }
    F evaluation_at_point(size_t, const F &, const std::vector<F> &c) const override
    {
        if (c.size() != num_oracles) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        F result = F::zero();
        for (size_t i = 0; i < c.size(); ++i) result += coefficients[i] * c[i];
        return result;
    }
};

// sumcheck.tcc:11-165
template<typename F>
struct sumcheck_g_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    D summation_domain, codeword_domain;
    F claimed_sum = F::zero();
    sumcheck_g_oracle(const D &H, const D &L) : summation_domain(H), codeword_domain(L) {}
    void set_claimed_sum(const F &mu) { claimed_sum = mu; }
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        if (c.size() != 2) throw std::invalid_argument("sumcheck_g_oracle has two constituent oracles");
        return sumcheck_g_contents<F>(*c[0], *c[1], codeword_domain, summation_domain, claimed_sum);
    }
    F at_point(const F &x, const F &f, const F &h, const affine_subspace<F> &H) const              // :137-145
    {
        const vanishing_polynomial<F, affine_subspace<F>> Z(H);
        const F eps_inv_mu = Z.lin[1].inverse() * claimed_sum;
        return f - eps_inv_mu * x.pow(H.num_elements() - 1) - Z.evaluation_at_point(x) * h;
    }
    F at_point(const F &x, const F &f, const F &h, const mult_coset<F> &H) const                   // :146-160
    {
        const vanishing_polynomial<F, mult_coset<F>> Z(H);
        return (f - F((uint64_t)H.order).inverse() * claimed_sum - Z.evaluation_at_point(x) * h) * x.inverse();
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override
    {
        if (c.size() != 2) throw std::invalid_argument("sumcheck_g_oracle has two constituent oracles");
        return at_point(x, c[0], c[1], summation_domain);
    }
};

// ldt_reducer_aux.tcc — the struct of ldt.hpp behind the virtual-oracle interface
template<typename F>
struct combined_LDT_oracle : virtual_oracle<F> {
    typedef domain_of<F> D;
    D codeword_domain;
    combined_LDT_virtual_oracle<F> inner;
    combined_LDT_oracle(const D &L, const std::vector<size_t> &degrees) : codeword_domain(L), inner(degrees) {}
    std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &c) const override
    {
        std::vector<std::vector<F>> evals;
        for (auto *p : c) evals.push_back(*p);
        return inner.evaluated_contents(codeword_domain, evals);
    }
    F evaluation_at_point(size_t, const F &x, const std::vector<F> &c) const override { return inner.evaluation_at_point(x, c); }
};

// ---- batch sumcheck (sumcheck.tcc:167-430), non-zk ----
template<typename F>
struct batch_sumcheck_protocol {
    typedef domain_of<F> D;
    bcs_protocol<F> &IOP;
    size_t summation_domain_handle, codeword_domain_handle, degree_bound;
    D summation_domain, codeword_domain;
    size_t g_degree, h_degree;
    std::vector<oracle_handle> oracle_handles;
    std::vector<F> claimed_sums;
    size_t challenge_handle = 0;
    oracle_handle h_handle{}, combined_f_handle{}, g_handle{};
    std::shared_ptr<random_linear_combination_oracle<F>> combined_f_oracle;
    std::shared_ptr<sumcheck_g_oracle<F>> g_oracle;

    batch_sumcheck_protocol(bcs_protocol<F> &iop, size_t summation_h, size_t codeword_h, size_t degree)
        : IOP(iop), summation_domain_handle(summation_h), codeword_domain_handle(codeword_h), degree_bound(degree),
          summation_domain(iop.get_domain(summation_h)), codeword_domain(iop.get_domain(codeword_h))
    {
        g_degree = dom_size(summation_domain) - 1;
        h_degree = degree_bound - dom_size(summation_domain);
    }
    void attach_oracle_for_summing(const oracle_handle &h, const F &claimed_sum = F::zero())
    {
        if (combined_f_oracle) throw std::logic_error("Called attach_oracle_for_summing after register_proof.");
        oracle_handles.push_back(h);
        claimed_sums.push_back(claimed_sum);
    }
    void register_challenge() { challenge_handle = IOP.register_verifier_random_message(oracle_handles.size()); }      // :199-206
    void register_proof()                                                                                              // :235-273
    {
        h_handle = IOP.register_oracle(codeword_domain_handle, h_degree, false);
        combined_f_oracle = std::make_shared<random_linear_combination_oracle<F>>(oracle_handles.size());
        combined_f_handle = IOP.register_virtual_oracle(codeword_domain_handle, degree_bound, oracle_handles, combined_f_oracle, true);
        g_oracle = std::make_shared<sumcheck_g_oracle<F>>(summation_domain, codeword_domain);
        g_handle = IOP.register_virtual_oracle(codeword_domain_handle, g_degree, { combined_f_handle, h_handle }, g_oracle);
    }
    F get_combined_claimed_sum(const std::vector<F> &challenge) const                                                  // :327-341
    {
        F s = F::zero();
        for (size_t i = 0; i < claimed_sums.size(); ++i) s += challenge[i] * claimed_sums[i];
        return s;
    }
    void calculate_and_submit_proof()                                                                                  // :343-388
    {
        const std::vector<F> challenge = IOP.obtain_verifier_random_message(challenge_handle);
        combined_f_oracle->set_random_coefficients(challenge);
        const std::vector<F> &evals = IOP.get_oracle_evaluations(combined_f_handle);
        std::vector<F> poly = IFFT_of_known_degree_over<F>(evals, degree_bound, codeword_domain);
        poly.resize(degree_bound);
        g_oracle->set_claimed_sum(get_combined_claimed_sum(challenge));
        const vanishing_polynomial<F, D> Z(summation_domain);
        std::vector<F> h = Z.divide(poly).first;
        IOP.submit_oracle(h_handle, FFT_over<F>(h, codeword_domain));
        IOP.drop_scratch();
    }
    void construct_verifier_state()                                                                                    // :391-401
    {
        const std::vector<F> challenge = IOP.obtain_verifier_random_message(challenge_handle);
        combined_f_oracle->set_random_coefficients(challenge);
        g_oracle->set_claimed_sum(get_combined_claimed_sum(challenge));
    }
    std::vector<oracle_handle> get_all_oracle_handles() const { return { h_handle, g_handle }; }                       // :403-414
};

// ---- multi lincheck (basic_lincheck.tcc:113-296) ----
template<typename F>
struct multi_lincheck {
    typedef domain_of<F> D;
    typedef std::vector<typename r1cs_system<F>::row> matrix;
    bcs_protocol<F> &IOP;
    size_t codeword_domain_handle, summation_domain_handle, num_matrices, repetitions, lincheck_degree;
    std::vector<oracle_handle> constituent_oracle_handles;
    std::vector<std::shared_ptr<batch_sumcheck_protocol<F>>> sumchecks;
    std::vector<std::shared_ptr<multi_lincheck_virtual_oracle<F>>> oracles;
    std::vector<size_t> alpha_handles, random_coefficient_handles;

    multi_lincheck(bcs_protocol<F> &iop, size_t codeword_h, size_t constraint_h, size_t variable_h, size_t input_variable_dim,
                   const std::vector<const matrix *> &matrices, const oracle_handle &fz_handle, const std::vector<oracle_handle> &Mz_handles,
                   size_t reps)
        : IOP(iop), codeword_domain_handle(codeword_h), num_matrices(matrices.size()), repetitions(reps)
    {
        if (num_matrices < 1) throw std::invalid_argument("multi_lincheck expects at least one matrix");
        if (Mz_handles.size() != num_matrices) throw std::invalid_argument("inconsistent number of Mz_handles and matrices passed into multi lincheck.");
        const D codeword_domain = IOP.get_domain(codeword_h), constraint_domain = IOP.get_domain(constraint_h), variable_domain = IOP.get_domain(variable_h);
        summation_domain_handle = dom_dim(constraint_domain) > dom_dim(variable_domain) ? constraint_h : variable_h;   // :137-143
        const D summation_domain = IOP.get_domain(summation_domain_handle);
        constituent_oracle_handles.push_back(fz_handle);
        for (auto &h : Mz_handles) constituent_oracle_handles.push_back(h);
        lincheck_degree = dom_size(summation_domain) + std::max(IOP.get_oracle_degree(fz_handle), IOP.get_oracle_degree(Mz_handles[0])) - 1;   // :151-154
        for (size_t i = 0; i < repetitions; ++i) {
            sumchecks.push_back(std::make_shared<batch_sumcheck_protocol<F>>(IOP, summation_domain_handle, codeword_h, lincheck_degree));
            oracles.push_back(std::make_shared<multi_lincheck_virtual_oracle<F>>(codeword_domain, constraint_domain, variable_domain, summation_domain,
                                                                                  input_variable_dim, matrices));
        }
    }
    void register_challenge()                                                                      // :197-218
    {
        for (size_t i = 0; i < repetitions; ++i) alpha_handles.push_back(IOP.register_verifier_random_message(1));
        for (size_t i = 0; i < repetitions; ++i) random_coefficient_handles.push_back(IOP.register_verifier_random_message(num_matrices));
        for (size_t i = 0; i < repetitions; ++i) {
            const oracle_handle h = IOP.register_virtual_oracle(codeword_domain_handle, lincheck_degree, constituent_oracle_handles, oracles[i]);
            sumchecks[i]->attach_oracle_for_summing(h);
            sumchecks[i]->register_challenge();
        }
    }
    void register_proof() { for (auto &s : sumchecks) s->register_proof(); }
    void set_challenges()
    {
        for (size_t i = 0; i < repetitions; ++i) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handles[i])[0];
            oracles[i]->set_challenge(alpha, IOP.obtain_verifier_random_message(random_coefficient_handles[i]));
        }
    }
    void calculate_and_submit_proof()                                                              // :241-257
    {
        for (size_t i = 0; i < repetitions; ++i) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handles[i])[0];
            oracles[i]->set_challenge(alpha, IOP.obtain_verifier_random_message(random_coefficient_handles[i]));
            sumchecks[i]->calculate_and_submit_proof();
        }
    }
    void construct_verifier_state()                                                                // :259-271
    {
        for (size_t i = 0; i < repetitions; ++i) {
            const F alpha = IOP.obtain_verifier_random_message(alpha_handles[i])[0];
            oracles[i]->set_challenge(alpha, IOP.obtain_verifier_random_message(random_coefficient_handles[i]));
            sumchecks[i]->construct_verifier_state();
        }
    }
    std::vector<oracle_handle> get_all_oracle_handles() const
    {
        std::vector<oracle_handle> out;
        for (auto &s : sumchecks) for (auto &h : s->get_all_oracle_handles()) out.push_back(h);
        return out;
    }
};

// ---- encoded Aurora (r1cs_rs_iop.tcc:252-693), non-zk, non-holographic ----
template<typename F>
struct encoded_aurora_protocol {
    typedef domain_of<F> D;
    bcs_protocol<F> &IOP;
    size_t constraint_domain_handle, variable_domain_handle, codeword_domain_handle;
    const r1cs_system<F> &cs;
    D constraint_domain, variable_domain, codeword_domain, input_variable_domain;
    oracle_handle fw_handle{}, fAz_handle{}, fBz_handle{}, fCz_handle{}, fz_handle{}, rowcheck_handle{};
    std::shared_ptr<fz_virtual_oracle<F>> fz_oracle;
    std::shared_ptr<rowcheck_ABC_virtual_oracle<F>> rowcheck_oracle;
    std::shared_ptr<multi_lincheck<F>> lincheck;

    // lincheck_repetitions == 0: the holographic arm (r1cs_rs_iop.tcc:344-357) — oracle/fractal.hpp attaches its own lincheck
    encoded_aurora_protocol(bcs_protocol<F> &iop, size_t constraint_h, size_t variable_h, size_t codeword_h, const r1cs_system<F> &system,
                            size_t lincheck_repetitions)
        : IOP(iop), constraint_domain_handle(constraint_h), variable_domain_handle(variable_h), codeword_domain_handle(codeword_h), cs(system),
          constraint_domain(iop.get_domain(constraint_h)), variable_domain(iop.get_domain(variable_h)), codeword_domain(iop.get_domain(codeword_h)),
          input_variable_domain(dom_subset_of_order(iop.get_domain(variable_h), system.num_inputs + 1))                 // :279-280
    {
        // register_witness_oracles (:285-375), query bound b = 0
        const size_t m = (size_t)1 << ceil_log2(cs.num_constraints()), n = (size_t)1 << ceil_log2(cs.num_variables), k = cs.num_inputs;
        const size_t fw_degree = n - (k + 1);
        fw_handle = IOP.register_oracle(codeword_h, fw_degree, false);
        fAz_handle = IOP.register_oracle(codeword_h, m, false);
        fBz_handle = IOP.register_oracle(codeword_h, m, false);
        fCz_handle = IOP.register_oracle(codeword_h, m, false);
        fz_oracle = std::make_shared<fz_virtual_oracle<F>>(k, input_variable_domain, codeword_domain);
        fz_handle = IOP.register_virtual_oracle(codeword_h, fw_degree + k + 1, { fw_handle }, fz_oracle);
        const std::vector<oracle_handle> Mz_handles = { fAz_handle, fBz_handle, fCz_handle };
        if (lincheck_repetitions)
            lincheck = std::make_shared<multi_lincheck<F>>(IOP, codeword_h, constraint_h, variable_h, dom_dim(input_variable_domain),
                                                           std::vector<const typename multi_lincheck<F>::matrix *>{ &cs.A, &cs.B, &cs.C }, fz_handle, Mz_handles,
                                                           lincheck_repetitions);
        rowcheck_oracle = std::make_shared<rowcheck_ABC_virtual_oracle<F>>(codeword_domain, constraint_domain);
        rowcheck_handle = IOP.register_virtual_oracle(codeword_h, dom_size(constraint_domain) - 1, Mz_handles, rowcheck_oracle);
    }
    void register_challenge() { lincheck->register_challenge(); }
    void register_proof() { lincheck->register_proof(); }

    void submit_witness_oracles(const std::vector<F> &primary_input, const std::vector<F> &auxiliary_input)             // :481-615
    {
        fz_oracle->set_primary_input(primary_input);
        const std::vector<F> f_1v_coefficients = fz_oracle->f_1v_coefficients();                                       // :508-516
        const std::vector<F> f_1v_over_variable_domain = FFT_over<F>(f_1v_coefficients, variable_domain);              // :517-518
        // create_fw_prime_evals (:406-430)
        std::vector<F> fw_prime_evals(dom_size(variable_domain), F::zero());
        const size_t input_variable_dim = ceil_log2(primary_input.size() + 1);
        for (size_t i = 0; i < auxiliary_input.size(); ++i) {
            const size_t variable_index = dom_reindex_by_subset(variable_domain, input_variable_dim, i + primary_input.size() + 1);
            fw_prime_evals[variable_index] = auxiliary_input[i] - f_1v_over_variable_domain[variable_index];
        }
        const std::vector<F> fw_prime = IFFT_over<F>(fw_prime_evals, variable_domain);                                 // :551-555
        const vanishing_polynomial<F, D> input_vp(input_variable_domain);
        const std::vector<F> fw = input_vp.divide(fw_prime).first;                                                     // :563-565
        std::vector<F> fw_over_codeword_domain = FFT_over<F>(fw, codeword_domain);                                     // :567-568
        std::vector<F> z = { F::one() };                                                                               // :581-592
        z.insert(z.end(), primary_input.begin(), primary_input.end());
        z.insert(z.end(), auxiliary_input.begin(), auxiliary_input.end());
        // compute_fprime_ABCz_over_codeword_domain (:432-479)
        std::vector<F> fAz = FFT_over<F>(IFFT_over<F>(sparse_times_vector<F>(cs.A, z), constraint_domain), codeword_domain);
        std::vector<F> fBz = FFT_over<F>(IFFT_over<F>(sparse_times_vector<F>(cs.B, z), constraint_domain), codeword_domain);
        std::vector<F> fCz = FFT_over<F>(IFFT_over<F>(sparse_times_vector<F>(cs.C, z), constraint_domain), codeword_domain);
        IOP.submit_oracle(fw_handle, std::move(fw_over_codeword_domain));                                              // :603-606
        IOP.submit_oracle(fAz_handle, std::move(fAz));
        IOP.submit_oracle(fBz_handle, std::move(fBz));
        IOP.submit_oracle(fCz_handle, std::move(fCz));
    }
    void calculate_and_submit_proof() { lincheck->calculate_and_submit_proof(); }
    void construct_verifier_state(const std::vector<F> &primary_input)
    {
        fz_oracle->set_primary_input(primary_input);
        lincheck->construct_verifier_state();
    }
    std::vector<oracle_handle> get_all_oracle_handles() const                                                          // :651-672
    {
        std::vector<oracle_handle> out = lincheck->get_all_oracle_handles();
        out.push_back(fw_handle); out.push_back(fAz_handle); out.push_back(fBz_handle); out.push_back(fCz_handle);
        out.push_back(rowcheck_handle);
        return out;
    }
};

// ---- FRI (fri_ldt.tcc:260-680) ----
template<typename F> std::vector<affine_subspace<F>> fri_domains(const affine_subspace<F> &L, const std::vector<size_t> &loc) { return fri_additive_domains<F>(L, loc); }
template<typename F> std::vector<mult_coset<F>> fri_domains(const mult_coset<F> &L, const std::vector<size_t> &loc)    // :292-308
{
    std::vector<mult_coset<F>> out = { L };
    size_t size = L.order;
    F shift = L.shift;
    for (size_t eta : loc) { shift = shift.pow((uint64_t)1 << eta); size >>= eta; out.push_back(mult_coset<F>(size, shift)); }
    return out;
}
template<typename F> std::vector<F> fri_fold(const std::vector<F> &f, const affine_subspace<F> &d, size_t cs, const F &x)
{
    timed_block tb("evaluating next FRI codeword");                                                   // fri_ldt.tcc:519
    return additive_evaluate_next_f_i_over_entire_domain<F>(f, d, cs, x);
}
template<typename F> std::vector<F> fri_fold(const std::vector<F> &f, const mult_coset<F> &d, size_t cs, const F &x)
{
    timed_block tb("evaluating next FRI codeword");
    return multiplicative_evaluate_next_f_i_over_entire_domain<F>(f, d, cs, x);
}
// evaluate_next_f_i_at_coset (fri_aux.tcc:251-349): `shift` is the queried coset's first element
template<typename F> F fri_fold_at_coset(const std::vector<F> &f, const affine_subspace<F> &d, size_t cs, const F &shift, const F &x)
{
    return additive_evaluate_next_f_i_at_coset<F>(f, d.subset_of_order(cs).basis, shift, x);
}
template<typename F> F fri_fold_at_coset(const std::vector<F> &f, const mult_coset<F> &, size_t cs, const F &shift, const F &x)
{
    return multiplicative_evaluate_next_f_i_at_coset<F>(f, F::subgroup_generator(cs), shift, x);
}
// localizer polynomial of the first 2^eta elements evaluated at a point (localizer_polynomial.tcc:3-35)
template<typename F> F fri_localize(const affine_subspace<F> &d, size_t cs, const F &x)
{
    return linearized_eval<F>(vanishing_polynomial_from_subspace<F>(affine_subspace<F>(d.subset_of_order(cs).basis, F::zero())), x);
}
template<typename F> F fri_localize(const mult_coset<F> &, size_t cs, const F &x) { return x.pow(cs); }

template<typename F>
struct FRI_protocol {
    typedef domain_of<F> D;
    struct query_set { position_handle s0; size_t interaction_index, LDT_index; std::vector<std::vector<size_t>> queries; };
    bcs_protocol<F> &IOP;
    size_t codeword_domain_handle;
    std::vector<oracle_handle> poly_handles;
    std::vector<size_t> localization;
    size_t poly_degree_bound, interactive_repetitions, query_repetitions, num_reductions, final_polynomial_degree_bound = 0;
    std::vector<D> domains;
    std::vector<size_t> domain_handles;
    std::vector<std::vector<std::vector<oracle_handle>>> oracle_handles;       // [reduction][interaction][LDT]
    std::vector<std::vector<size_t>> verifier_challenge_handles;              // [reduction][interaction]
    std::vector<std::vector<size_t>> final_polynomial_handles;                // [interaction][LDT]
    std::vector<query_set> query_sets;

    FRI_protocol(bcs_protocol<F> &iop, size_t codeword_h, const std::vector<oracle_handle> &polys, const std::vector<size_t> &loc,
                 size_t degree_bound, size_t interactions, size_t queries)
        : IOP(iop), codeword_domain_handle(codeword_h), poly_handles(polys), localization(loc), poly_degree_bound(degree_bound),
          interactive_repetitions(interactions), query_repetitions(queries), num_reductions(loc.size()),
          domains(fri_domains<F>(iop.get_domain(codeword_h), loc)) {}

    void register_interactions()                                                                   // :342-398
    {
        size_t total = localization[0];
        domain_handles.assign(num_reductions, 0);
        oracle_handles.resize(num_reductions);
        verifier_challenge_handles.resize(num_reductions);
        domain_handles[0] = codeword_domain_handle;
        oracle_handles[0] = { poly_handles };
        for (size_t j = 0; j < interactive_repetitions; ++j) verifier_challenge_handles[0].push_back(IOP.register_verifier_random_message(1));
        for (size_t i = 1; i < num_reductions; ++i) {
            total += localization[i];
            const size_t degree_bound = poly_degree_bound >> total;
            const size_t L_i = IOP.register_domain(domains[i]);
            for (size_t j = 0; j < interactive_repetitions; ++j) {
                std::vector<oracle_handle> multi_f_i;
                for (size_t l = 0; l < poly_handles.size(); ++l) multi_f_i.push_back(IOP.register_oracle(L_i, degree_bound, false));
                oracle_handles[i].push_back(multi_f_i);
            }
            IOP.set_round_parameters((size_t)1 << localization[i]);
            for (size_t j = 0; j < interactive_repetitions; ++j) verifier_challenge_handles[i].push_back(IOP.register_verifier_random_message(1));
            domain_handles[i] = L_i;
        }
        final_polynomial_degree_bound = poly_degree_bound >> total;
        for (size_t j = 0; j < interactive_repetitions; ++j) {
            std::vector<size_t> hs;
            for (size_t l = 0; l < poly_handles.size(); ++l) hs.push_back(IOP.register_prover_message(final_polynomial_degree_bound));
            final_polynomial_handles.push_back(hs);
        }
    }
    void register_queries()                                                                        // :400-472
    {
        for (size_t q = 0; q < query_repetitions; ++q) {
            const position_handle s0 = IOP.register_random_query_position(domain_handles[0]);
            std::vector<std::vector<position_handle>> coset_positions(num_reductions);
            {   // query_position_to_queries_for_entire_coset (iop/utilities/query_positions.tcc)
                const D d = domains[0];
                const size_t cs = (size_t)1 << localization[0];
                for (size_t i = 0; i < cs; ++i)
                    coset_positions[0].push_back(IOP.register_deterministic_query_position({ s0 }, [d, cs, i](const std::vector<size_t> &seed) {
                        return dom_position(d, dom_coset_index(d, seed[0], cs), i, cs);
                    }));
            }
            for (size_t r = 1; r < num_reductions; ++r) {                                          // calculate_next_coset_query_positions (fri_aux.tcc:351-387)
                const D prev = domains[r - 1], cur = domains[r];
                const size_t prev_cs = (size_t)1 << localization[r - 1], cur_cs = (size_t)1 << localization[r];
                for (size_t i = 0; i < cur_cs; ++i)
                    coset_positions[r].push_back(IOP.register_deterministic_query_position({ coset_positions[r - 1][0] },
                        [prev, cur, prev_cs, cur_cs, i](const std::vector<size_t> &seed) {
                            const size_t localized_position = dom_coset_index(prev, seed[0], prev_cs);
                            return dom_position(cur, dom_coset_index(cur, localized_position, cur_cs), i, cur_cs);
                        }));
            }
            for (size_t interaction = 0; interaction < interactive_repetitions; ++interaction)
                for (size_t ldt = 0; ldt < poly_handles.size(); ++ldt) {
                    query_set Q{ s0, interaction, ldt, {} };
                    Q.queries.resize(num_reductions);
                    for (size_t r = 0; r < num_reductions; ++r) {
                        const size_t queried_interaction = r == 0 ? 0 : interaction;
                        for (size_t j = 0; j < ((size_t)1 << localization[r]); ++j)
                            Q.queries[r].push_back(IOP.register_query(oracle_handles[r][queried_interaction][ldt], coset_positions[r][j]));
                    }
                    query_sets.push_back(Q);
                }
        }
    }
    void calculate_and_submit_proof()                                                              // :474-548
    {
        std::vector<std::vector<F>> first;
        for (auto &h : poly_handles) first.push_back(IOP.get_oracle_evaluations(h));
        IOP.drop_scratch();
        std::vector<std::vector<std::vector<F>>> by_interaction(interactive_repetitions, first);
        for (size_t i = 0; i < num_reductions; ++i) {
            const size_t cs = (size_t)1 << localization[i];
            if (i > 0) {
                for (size_t j = 0; j < interactive_repetitions; ++j)
                    for (size_t l = 0; l < poly_handles.size(); ++l) IOP.submit_oracle(oracle_handles[i][j][l], std::vector<F>(by_interaction[j][l]));
                IOP.signal_prover_round_done();
            }
            for (size_t j = 0; j < interactive_repetitions; ++j) {
                const F x_i = IOP.obtain_verifier_random_message(verifier_challenge_handles[i][j])[0];
                for (size_t l = 0; l < poly_handles.size(); ++l) by_interaction[j][l] = fri_fold<F>(by_interaction[j][l], domains[i], cs, x_i);
            }
        }
        for (size_t j = 0; j < interactive_repetitions; ++j)
            for (size_t l = 0; l < poly_handles.size(); ++l) {
                std::vector<F> coeffs = IFFT_over<F>(by_interaction[j][l], domains[num_reductions]);
                coeffs.resize(final_polynomial_degree_bound);
                IOP.submit_prover_message(final_polynomial_handles[j][l], std::move(coeffs));
            }
        IOP.signal_prover_round_done();
    }
    bool predicate_for_query_set(const query_set &Q)                                               // :573-651
    {
        const size_t s0_idx = IOP.obtain_query_position(Q.s0);
        F si = dom_element(domains[0], s0_idx);
        size_t si_idx = s0_idx;
        F last_interpolation = F::zero();
        for (size_t i = 0; i < num_reductions; ++i) {
            const F x_i = IOP.obtain_verifier_random_message(verifier_challenge_handles[i][Q.interaction_index])[0];
            const size_t cs = (size_t)1 << localization[i];
            const size_t si_j = dom_coset_index(domains[i], si_idx, cs), si_k = dom_intra_coset_index(domains[i], si_idx, cs);
            si_idx = si_j;
            std::vector<F> fi_on_coset;
            for (size_t k = 0; k < cs; ++k) fi_on_coset.push_back(IOP.obtain_query_response(Q.queries[i][k]));
            if (i > 0 && last_interpolation != fi_on_coset[si_k]) return false;
            const F shift = dom_element(domains[i], dom_position(domains[i], si_j, 0, cs));
            last_interpolation = fri_fold_at_coset<F>(fi_on_coset, domains[i], cs, shift, x_i);
            si = fri_localize<F>(domains[i], cs, si);
        }
        const std::vector<F> last_poly = IOP.receive_prover_message(final_polynomial_handles[Q.interaction_index][Q.LDT_index]);
        return poly_eval<F>(last_poly, si) == last_interpolation;
    }
    bool verifier_predicate()                                                                      // :550-571
    {
        bool decision = true;
        for (auto &Q : query_sets) if (!predicate_for_query_set(Q)) decision = false;
        return decision;
    }
};

// ---- LDT instance reducer (ldt_reducer.tcc:134-297), non-zk ----
template<typename F>
struct LDT_instance_reducer {
    typedef domain_of<F> D;
    bcs_protocol<F> &IOP;
    size_t codeword_domain_handle, num_output_LDT_instances, max_tested_degree_bound;
    std::vector<oracle_handle> input_oracle_handles, combined_oracle_handles;
    std::vector<std::shared_ptr<combined_LDT_oracle<F>>> combined_oracles;
    std::vector<size_t> random_coefficients_handles;
    std::shared_ptr<FRI_protocol<F>> multi_LDT;

    LDT_instance_reducer(bcs_protocol<F> &iop, size_t codeword_h, size_t instances, size_t max_tested)
        : IOP(iop), codeword_domain_handle(codeword_h), num_output_LDT_instances(instances), max_tested_degree_bound(max_tested) {}

    void register_interactions(const std::vector<oracle_handle> &handles, const std::vector<size_t> &localization, size_t fri_interactions, size_t fri_queries)
    {
        input_oracle_handles = handles;
        std::vector<size_t> degrees;
        for (auto &h : handles) {
            degrees.push_back(IOP.get_oracle_degree(h));
            if (degrees.back() > max_tested_degree_bound)
                throw std::invalid_argument("One of the oracles is registered with claimed degree greater than the max tested degree bound");
        }
        const size_t num_random_coefficients = 2 * handles.size();
        for (size_t i = 0; i < num_output_LDT_instances; ++i) {
            combined_oracles.push_back(std::make_shared<combined_LDT_oracle<F>>(IOP.get_domain(codeword_domain_handle), degrees));
            combined_oracle_handles.push_back(IOP.register_virtual_oracle(codeword_domain_handle, max_tested_degree_bound, handles, combined_oracles[i]));
        }
        for (size_t i = 0; i < num_output_LDT_instances; ++i) random_coefficients_handles.push_back(IOP.register_verifier_random_message(num_random_coefficients));
        multi_LDT = std::make_shared<FRI_protocol<F>>(IOP, codeword_domain_handle, combined_oracle_handles, localization, max_tested_degree_bound,
                                                      fri_interactions, fri_queries);
        multi_LDT->register_interactions();
    }
    void register_queries() { multi_LDT->register_queries(); }
    void set_coefficients()
    {
        for (size_t i = 0; i < num_output_LDT_instances; ++i) combined_oracles[i]->inner.set_random_coefficients(IOP.obtain_verifier_random_message(random_coefficients_handles[i]));
    }
    void calculate_and_submit_proof() { set_coefficients(); multi_LDT->calculate_and_submit_proof(); }
    bool verifier_predicate() { set_coefficients(); return multi_LDT->verifier_predicate(); }
};

// ---- aurora_iop (aurora_iop.tcc:262-359) ----
template<typename F>
struct aurora_iop {
    typedef domain_of<F> D;
    bcs_protocol<F> &IOP;
    const aurora_parameters<F> &params;
    size_t codeword_domain_handle;
    std::shared_ptr<encoded_aurora_protocol<F>> protocol;
    std::shared_ptr<LDT_instance_reducer<F>> LDT_reducer;

    aurora_iop(bcs_protocol<F> &iop, const r1cs_system<F> &cs, const aurora_parameters<F> &p) : IOP(iop), params(p)
    {
        const D unshifted = default_domain<F>((size_t)1 << p.codeword_domain_dim);
        const F codeword_domain_shift = dom_element_outside(unshifted);                            // :282-283
        const size_t constraint_h = IOP.register_domain(default_domain<F>((size_t)1 << p.constraint_domain_dim));
        const size_t variable_h = IOP.register_domain(default_domain<F>((size_t)1 << p.variable_domain_dim));
        codeword_domain_handle = IOP.register_domain(shifted_domain<F>((size_t)1 << p.codeword_domain_dim, codeword_domain_shift));
        protocol = std::make_shared<encoded_aurora_protocol<F>>(IOP, constraint_h, variable_h, codeword_domain_handle, cs, p.multi_lincheck_repetitions);
        LDT_reducer = std::make_shared<LDT_instance_reducer<F>>(IOP, codeword_domain_handle, p.num_output_LDT_instances, p.max_tested_degree_bound);
        IOP.set_round_parameters((size_t)1 << p.localization_parameters[0]);                       // :307-308
    }
    void register_interactions()                                                                   // :311-326
    {
        protocol->register_challenge();
        protocol->register_proof();
        IOP.set_round_parameters((size_t)1 << params.localization_parameters[0]);
        LDT_reducer->register_interactions(protocol->get_all_oracle_handles(), params.localization_parameters, params.fri_interactive_repetitions,
                                           params.fri_query_repetitions);
    }
    void register_queries() { LDT_reducer->register_queries(); }
    void produce_proof(const std::vector<F> &primary_input, const std::vector<F> &auxiliary_input)  // :334-344
    {
        protocol->submit_witness_oracles(primary_input, auxiliary_input);
        IOP.signal_prover_round_done();
        protocol->calculate_and_submit_proof();
        IOP.signal_prover_round_done();
        LDT_reducer->calculate_and_submit_proof();
    }
    bool verifier_predicate(const std::vector<F> &primary_input)                                   // :346-357
    {
        protocol->construct_verifier_state(primary_input);
        return LDT_reducer->verifier_predicate();
    }
};

// aurora_snark.tcc:119-146
template<typename F>
bcs_transcript<F> aurora_snark_prover(const r1cs_system<F> &cs, const std::vector<F> &primary_input, const std::vector<F> &auxiliary_input,
                                      const aurora_parameters<F> &params)
{
    timed_block tb("Aurora SNARK prover");                                                            // aurora_snark.tcc:126
    bcs_protocol<F> IOP(params.pow_bits);
    aurora_iop<F> full_protocol(IOP, cs, params);
    full_protocol.register_interactions();
    IOP.seal_interaction_registrations();
    full_protocol.register_queries();
    IOP.seal_query_registrations();
    full_protocol.produce_proof(primary_input, auxiliary_input);
    timed_block tb_transcript("Obtain transcript");                                                   // aurora_snark.tcc:138
    return IOP.get_transcript();
}

// aurora_snark.tcc:148-186
template<typename F>
bool aurora_snark_verifier(const r1cs_system<F> &cs, const std::vector<F> &primary_input, const bcs_transcript<F> &proof, const aurora_parameters<F> &params)
{
    try {
        bcs_protocol<F> IOP(params.pow_bits, proof);
        aurora_iop<F> full_protocol(IOP, cs, params);
        full_protocol.register_interactions();
        IOP.seal_interaction_registrations();
        full_protocol.register_queries();
        IOP.seal_query_registrations();
        const bool valid = IOP.transcript_is_valid();
        if (!valid) return false;
        return full_protocol.verifier_predicate(primary_input);
    } catch (const std::exception &) {
        return false;                       // malformed transcripts surface as exceptions in the reference
    }
}


// ---- FRI-only SNARK (config 3): libiop/snark/fri_snark.tcc:21-126 over libiop/protocols/fri_iop.tcc ----
// Reference quirk F14: dummy_oracle::evaluated_contents (protocols/encoded/dummy_protocol.tcc:14-32) reserves its result and then
// loops over its (still zero) size, so the virtual oracle the reference hands to the LDT reducer is EMPTY and FRI_snark_prover
// folds out-of-bounds memory; no reference test runs it.  BASELINE config 3 states the intent — the FRI prover on a degree-2^20
// Reed-Solomon codeword — so the LDT reducer here takes the submitted oracle itself (one input of maximal degree: the combined
// oracle is that codeword).  Everything else follows fri_iop.tcc: unshifted default codeword domain (:13), one LDT instance, the
// given interactive / query repetitions (:71-73), round-0 leaves of 2^eta_0 (:55-57), pow parameter dim + 3 (fri_snark.tcc:26-28).
template<typename F>
struct FRI_snark_parameters {
    size_t codeword_domain_dim, RS_extra_dimensions, num_interactive_repetitions, num_query_repetitions;
    std::vector<size_t> localization_parameters;
    size_t pow_bits() const { return codeword_domain_dim + 3; }
    size_t poly_degree_bound() const { return (size_t)1 << (codeword_domain_dim - RS_extra_dimensions); }
};

template<typename F>
struct FRI_iop_protocol {
    bcs_protocol<F> &IOP;
    const FRI_snark_parameters<F> &params;
    size_t codeword_domain_handle;
    oracle_handle oracle{};
    std::shared_ptr<LDT_instance_reducer<F>> LDT;
    FRI_iop_protocol(bcs_protocol<F> &iop, const FRI_snark_parameters<F> &p) : IOP(iop), params(p)
    {
        codeword_domain_handle = IOP.register_domain(default_domain<F>((size_t)1 << p.codeword_domain_dim));
        oracle = IOP.register_oracle(codeword_domain_handle, p.poly_degree_bound(), false);
        LDT = std::make_shared<LDT_instance_reducer<F>>(IOP, codeword_domain_handle, 1, p.poly_degree_bound());
        IOP.set_round_parameters((size_t)1 << p.localization_parameters[0]);
    }
    void register_interactions()
    {
        LDT->register_interactions({ oracle }, params.localization_parameters, params.num_interactive_repetitions, params.num_query_repetitions);
    }
    void register_queries() { LDT->register_queries(); }
    void produce_proof(const std::vector<F> &poly_coeffs)                                         // fri_iop.tcc:82-89
    {
        IOP.submit_oracle(oracle, FFT_over<F>(poly_coeffs, IOP.get_domain(codeword_domain_handle)));
        IOP.signal_prover_round_done();
        LDT->calculate_and_submit_proof();
    }
};

template<typename F>
bcs_transcript<F> FRI_snark_prover(const std::vector<F> &poly_coeffs, const FRI_snark_parameters<F> &params)
{
    bcs_protocol<F> IOP(params.pow_bits());
    FRI_iop_protocol<F> full_protocol(IOP, params);
    full_protocol.register_interactions();
    IOP.seal_interaction_registrations();
    full_protocol.register_queries();
    IOP.seal_query_registrations();
    full_protocol.produce_proof(poly_coeffs);
    return IOP.get_transcript();
}

template<typename F>
bool FRI_snark_verifier(const bcs_transcript<F> &proof, const FRI_snark_parameters<F> &params)
{
    try {
        bcs_protocol<F> IOP(params.pow_bits(), proof);
        FRI_iop_protocol<F> full_protocol(IOP, params);
        full_protocol.register_interactions();
        IOP.seal_interaction_registrations();
        full_protocol.register_queries();
        IOP.seal_query_registrations();
        if (!IOP.transcript_is_valid()) return false;
        return full_protocol.LDT->verifier_predicate();
    } catch (const std::exception &) {
        return false;
    }
}

} // namespace oracle
