// Fractal preprocessing SNARK (non-zk, BLAKE2b) for C++ callers: indexer and prover, every vector in HBM.
//
//   fractal_snark_indexer / fractal_snark_prover / fractal_snark_parameters   libiop/snark/fractal_snark.tcc:7-162
//   fractal_iop_parameters, fractal_iop                                       libiop/protocols/fractal_hiop.tcc:5-329
//   matrix_indexer                                                            libiop/protocols/encoded/r1cs_rs_iop/fractal_indexer.tcc
//   holographic_multi_lincheck (+ its virtual oracle, single_matrix_denominator)   .../encoded/lincheck/holographic_lincheck{,_aux}.tcc
//   compute_p_alpha_M                                                         .../encoded/lincheck/common.tcc
//   rational_sumcheck_protocol, sumcheck_constraint_oracle                    .../encoded/sumcheck/rational_sumcheck.tcc
//   rational_linear_combination, single_boundary_constraint                   .../encoded/common/
//   bcs_indexer, bcs_prover's index handling                                  libiop/bcs/bcs_indexer.tcc, bcs_prover.tcc:12-21,68-80,119-134
//
// The encoded witness part (f_w, f_Az, f_Bz, f_Cz, fz, rowcheck), the batch sumcheck over H, the LDT instance reducer and FRI are
// aurora.hpp's.  Citations are relative to the reference tree.
#pragma once
#include "aurora.hpp"

namespace libiop_amd {

namespace dev {

template<typename FieldT>
device_vector<FieldT> mul(const device_vector<FieldT> &a, const device_vector<FieldT> &b)
{
    if (a.size() != b.size()) throw std::invalid_argument("mul: size mismatch");
    device_vector<FieldT> out(a.size());
    auto fn = field_host<FieldT>::additive() ? iopx_gf192_mul_dev : iopx_fp3_mul_dev;
    check(fn(a.words(), b.words(), out.words(), a.size()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> domain_offsets(const field_subset<FieldT> &D_in, const FieldT &point)          // point - x over the domain (this rank's part of it)
{
    const field_subset<FieldT> D = dist::local_domain(D_in);
    device_vector<FieldT> out(D.num_elements());
    if (additive(D)) check(iopx_domain_offsets_gf192_dev(basis_words(D), D.dimension(), shift_words(D), detail::words(&point), out.words()));
    else check(iopx_domain_offsets_fp3_dev(D.dimension(), gen_words(D), shift_words(D), detail::words(&point), out.words()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> domain_elements(const field_subset<FieldT> &D)                                 // field_subset::all_elements on the device
{
    if (additive(D)) return domain_offsets<FieldT>(D, field_host<FieldT>::zero());
    return pow_table<FieldT>(D.num_elements(), D.generator(), D.shift());
}

// lagrange_polynomial(x, S, normalized = false).evaluations_over_field_subset(evaldomain) (lagrange_polynomial.tcc:66-136):
// (Z_S(x) - Z_S(y)) / (x - y) for y over evaldomain.  The reference patches the position y = x (probability |evaldomain| / |F| for a
// sampled x) with the formal derivative; that case is refused here instead of silently differing.
template<typename FieldT>
device_vector<FieldT> lagrange_evals(const FieldT &x, const field_subset<FieldT> &S, const field_subset<FieldT> &evaldomain)
{
    if (field_host<FieldT>::element_in_domain(evaldomain, x)) throw std::logic_error("the evaluation point lies in the evaluation domain");
    const device_vector<FieldT> numerator = vanishing_evals<FieldT>(S, evaldomain, field_host<FieldT>::vanishing_eval(S, x));
    return div<FieldT>(&numerator, domain_offsets<FieldT>(evaldomain, x));
}

template<typename FieldT>
device_vector<FieldT> lincomb_affine(const std::vector<device_vector<FieldT>> &oracles, const std::vector<FieldT> &coefficients, const FieldT &constant)
{
    const std::vector<const void *> ptrs = pointers(oracles);
    device_vector<FieldT> out(oracles[0].size());
    auto fn = field_host<FieldT>::additive() ? iopx_lincomb_affine_gf192_dev : iopx_lincomb_affine_fp3_dev;
    check(fn(ptrs.data(), ptrs.size(), detail::words(coefficients.data()), detail::words(&constant), oracles[0].size(), out.words()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> scaled(const device_vector<FieldT> &v, const FieldT &scale)
{
    const void *ptr = v.data();
    device_vector<FieldT> out(v.size());
    auto fn = field_host<FieldT>::additive() ? iopx_lincomb_gf192_dev : iopx_lincomb_fp3_dev;
    check(fn(&ptr, 1, detail::words(&scale), v.size(), out.words()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> gathered(const device_vector<FieldT> &src, const std::vector<uint64_t> &index)
{
    const device_array<uint64_t> d_idx = device_array<uint64_t>::from_host(index);
    device_vector<FieldT> out(index.size());
    check(iopx_gather_dev(src.data(), d_idx.data(), index.size(), sizeof(FieldT), out.data()));
    return out;
}

} // namespace dev

// ---- parameters (fractal_snark.tcc:7-112, fractal_hiop.tcc:5-216; non-zk, heuristic FRI soundness, optimistic-heuristic LDT-reducer
// soundness; profiling/instrument_fractal_snark.cpp:93-120: RS_extra_dimensions 3, localization 2) ----
template<typename FieldT>
struct fractal_snark_parameters {
    std::size_t security_parameter_, RS_extra_dimensions_, num_constraints_, num_variables_, num_inputs_;
    std::size_t index_domain_dim_, matrix_domain_dim_, codeword_domain_dim_, pow_bits_, query_soundness_error_bits_, interactive_soundness_error_bits_;
    std::vector<std::size_t> localization_parameters_;
    std::size_t holographic_lincheck_repetitions_, max_tested_degree_bound_, max_constraint_degree_bound_, max_LDT_tested_degree_bound_;
    std::size_t absolute_proximity_parameter_, num_output_LDT_instances_, fri_query_repetitions_, fri_interactive_repetitions_;

    fractal_snark_parameters(const r1cs_constraint_system<FieldT> &cs, std::size_t security_parameter = 128, std::size_t RS_extra_dimensions = 3,
                             std::size_t FRI_localization_parameter = 2)
        : security_parameter_(security_parameter), RS_extra_dimensions_(RS_extra_dimensions)
    {
        typedef aurora_snark_parameters<FieldT> A;
        const std::size_t n = cs.num_constraints();
        if (security_parameter != 128) throw std::invalid_argument("libiop_amd: security_parameter must be 128 (32-byte BLAKE2b digests)");
        if (!A::is_pow2(n)) throw std::invalid_argument("Fractal requires the number of constraints to be a power of two");
        if (n != cs.num_variables() + 1) throw std::invalid_argument("Fractal requires the matrices to be square");
        num_constraints_ = n; num_variables_ = cs.num_variables(); num_inputs_ = cs.num_inputs();
        const std::size_t max_nonzero = std::max((std::size_t)cs.A.row_ptr.back(), std::max((std::size_t)cs.B.row_ptr.back(), (std::size_t)cs.C.row_ptr.back()));   // fractal_hiop.tcc:28-35
        index_domain_dim_ = detail::log2_ceil(max_nonzero);
        matrix_domain_dim_ = detail::log2_ceil(n);
        codeword_domain_dim_ = detail::log2_ceil((std::size_t)4 << index_domain_dim_) + RS_extra_dimensions;       // :38-39
        pow_bits_ = detail::log2_ceil(n) + 3;                                                 // fractal_snark.tcc:90-95
        query_soundness_error_bits_ = security_parameter + 1 - pow_bits_;                     // fractal_hiop.tcc:77-78
        interactive_soundness_error_bits_ = security_parameter + 3;
        localization_parameters_ = localization_parameter_to_array(FRI_localization_parameter, codeword_domain_dim_, RS_extra_dimensions);
        const double fbits = (double)field_host<FieldT>::soundness_bits();
        holographic_lincheck_repetitions_ = A::repetitions((double)interactive_soundness_error_bits_, 1.0 + (double)matrix_domain_dim_ - fbits);   // holographic_lincheck.tcc:16-36
        const std::size_t H = (std::size_t)1 << matrix_domain_dim_;
        max_tested_degree_bound_ = std::max(3 * H, H - 1);                                    // r1cs_rs_iop.tcc:56-97, holographic, b = 0
        max_constraint_degree_bound_ = std::max(4 * H, 2 * H - 1);
        std::size_t total_localization = 0;
        for (std::size_t l : localization_parameters_) total_localization += l;
        const std::size_t step = (std::size_t)1 << total_localization, rem = max_tested_degree_bound_ % step;      // next_testable_degree_bound (fri_ldt.tcc:148-163)
        max_LDT_tested_degree_bound_ = rem == 0 ? max_tested_degree_bound_ : max_tested_degree_bound_ - rem + step;
        const std::size_t codeword_size = (std::size_t)1 << codeword_domain_dim_;
        if (max_LDT_tested_degree_bound_ >= codeword_size || max_constraint_degree_bound_ >= codeword_size) throw std::invalid_argument("degree bounds exceed the codeword domain");
        absolute_proximity_parameter_ = std::min(codeword_size - max_constraint_degree_bound_, codeword_size - max_LDT_tested_degree_bound_) - 1;
        num_output_LDT_instances_ = A::repetitions((double)interactive_soundness_error_bits_, (double)codeword_domain_dim_ - fbits);
        const double delta = (double)absolute_proximity_parameter_ / (double)codeword_size;
        fri_query_repetitions_ = A::repetitions((double)query_soundness_error_bits_, std::log2(1 - delta));
        const double per_interaction = std::log2((double)(((std::size_t)1 << localization_parameters_[0]) - 1)) - fbits;
        fri_interactive_repetitions_ = A::repetitions((double)interactive_soundness_error_bits_, per_interaction);
    }
};

// ---- virtual oracles ------------------------------------------------------------------------------------------------------------
template<typename FieldT>
class holographic_multi_lincheck_virtual_oracle : public virtual_oracle<FieldT> {           // holographic_lincheck_aux.tcc:4-95
    field_subset<FieldT> codeword_domain_, summation_domain_;
    std::size_t num_matrices_;
    FieldT alpha_;
    std::vector<FieldT> r_Mz_;
public:
    holographic_multi_lincheck_virtual_oracle(const field_subset<FieldT> &L, const field_subset<FieldT> &H, std::size_t num_matrices)
        : codeword_domain_(L), summation_domain_(H), num_matrices_(num_matrices), alpha_(field_host<FieldT>::zero()) {}
    void set_challenge(const FieldT &alpha, const std::vector<FieldT> &r_Mz)
    {
        if (r_Mz.size() != num_matrices_) throw std::invalid_argument("Not enough random linear combination coefficients were provided");
        alpha_ = alpha; r_Mz_ = r_Mz;
    }
    // p(alpha, x) sum_m r_m f_Mz(x) - f_z(x) t(x); constituents (fz, Mz..., t)
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    std::size_t smallest_window() const override { return summation_domain_.num_elements(); }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != num_matrices_ + 2) throw std::invalid_argument("multi_lincheck uses more constituent oracles than what was provided.");
        const device_vector<FieldT> p_alpha_prime = dev::lagrange_evals<FieldT>(alpha_, summation_domain_, D);                      // :37-39
        std::vector<const void *> Mz;
        for (std::size_t m = 0; m < num_matrices_; ++m) Mz.push_back(c[1 + m].data());
        device_vector<FieldT> out(c[0].size());
        auto fn = field_host<FieldT>::additive() ? iopx_lincheck_gf192_dev : iopx_lincheck_fp3_dev;
        check(fn(c[0].words(), Mz.data(), Mz.size(), detail::words(r_Mz_.data()), p_alpha_prime.words(), c.back().words(), c[0].size(), out.words()));
        return out;
    }
};

template<typename FieldT>
class single_matrix_denominator : public virtual_oracle<FieldT> {                            // holographic_lincheck_aux.tcc:97-169
    FieldT row_query_point_, column_query_point_;
public:
    single_matrix_denominator() : row_query_point_(field_host<FieldT>::zero()), column_query_point_(field_host<FieldT>::zero()) {}
    void set_challenge(const FieldT &row_query_point, const FieldT &column_query_point) { row_query_point_ = row_query_point; column_query_point_ = column_query_point; }
    bool restrictable() const override { return true; }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &, const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents(c); }
    // (row - row_query)(col - col_query) from (row, col, row*col)
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override
    {
        typedef field_host<FieldT> H;
        if (c.size() != 3) throw std::invalid_argument("single_matrix_denominator was expecting row, col, row*col oracles as input");
        return dev::lincomb_affine<FieldT>(c, { H::neg(column_query_point_), H::neg(row_query_point_), H::one() }, H::mul(row_query_point_, column_query_point_));
    }
};

// rational_linear_combination (common/rational_linear_combination.tcc:136-212): registers the combined numerator and the combined
// denominator as two virtual oracles; one kernel produces both, so the pair is computed on the first request and the second is
// served from it
template<typename FieldT>
class rational_linear_combination {
    struct shared_state {
        std::size_t num_rationals;
        std::vector<FieldT> coefficients;
        std::vector<const void *> last_key;
        device_vector<FieldT> last_N, last_D;
        std::vector<device_vector<FieldT>> last_denominators;      // keeps the keyed buffers alive
        std::pair<device_vector<FieldT>, device_vector<FieldT>> pair(const std::vector<device_vector<FieldT>> &numerators, const std::vector<device_vector<FieldT>> &denominators)
        {
            const std::vector<const void *> np = dev::pointers(numerators), dp = dev::pointers(denominators);
            device_vector<FieldT> N(numerators[0].size()), D(numerators[0].size());
            auto fn = field_host<FieldT>::additive() ? iopx_rational_combine_gf192_dev : iopx_rational_combine_fp3_dev;
            check(fn(np.data(), dp.data(), num_rationals, detail::words(coefficients.data()), numerators[0].size(), N.words(), D.words()));
            last_key = dp; last_N = N; last_D = D; last_denominators = denominators;
            return { N, D };
        }
    };
    struct numerator_oracle : virtual_oracle<FieldT> {
        std::shared_ptr<shared_state> st;
        bool restrictable() const override { return true; }
        device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &, const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents(c); }
        device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override
        {
            const std::size_t n = st->num_rationals;
            if (c.size() != 2 * n) throw std::invalid_argument("Expected same number of evaluations as in registration.");
            return st->pair(std::vector<device_vector<FieldT>>(c.begin(), c.begin() + n), std::vector<device_vector<FieldT>>(c.begin() + n, c.end())).first;
        }
    };
    struct denominator_oracle : virtual_oracle<FieldT> {
        std::shared_ptr<shared_state> st;
        bool restrictable() const override { return true; }
        device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &, const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents(c); }
        device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override
        {
            if (c.size() != st->num_rationals) throw std::invalid_argument("Expected same number of evaluations as in registration.");
            if (!st->last_key.empty() && st->last_key == dev::pointers(c)) return st->last_D;
            device_vector<FieldT> out = c[0];
            for (std::size_t i = 1; i < c.size(); ++i) out = dev::mul<FieldT>(out, c[i]);
            return out;
        }
    };
    std::shared_ptr<shared_state> st_;
    oracle_handle numerator_handle_, denominator_handle_;
public:
    rational_linear_combination(bcs_prover<FieldT> &IOP, std::size_t num_rationals, const std::vector<oracle_handle> &numerator_handles,
                                const std::vector<oracle_handle> &denominator_handles)
        : st_(std::make_shared<shared_state>())
    {
        if (numerator_handles.size() != num_rationals || denominator_handles.size() != num_rationals)
            throw std::invalid_argument("Rational Linear Combination: #numerator handles passed in != #denominator handles passed in");
        st_->num_rationals = num_rationals;
        const domain_handle domain = IOP.get_oracle_domain(numerator_handles[0]);
        std::size_t denominator_degree = 1;
        for (auto &h : denominator_handles) denominator_degree += IOP.get_oracle_degree(h) - 1;
        auto den = std::make_shared<denominator_oracle>();
        den->st = st_;
        denominator_handle_ = IOP.register_virtual_oracle(domain, denominator_degree, denominator_handles, den);
        std::size_t numerator_degree = 0;
        for (std::size_t i = 0; i < num_rationals; ++i)
            numerator_degree = std::max(numerator_degree, IOP.get_oracle_degree(numerator_handles[i]) + denominator_degree - IOP.get_oracle_degree(denominator_handles[i]));
        std::vector<oracle_handle> all = numerator_handles;
        all.insert(all.end(), denominator_handles.begin(), denominator_handles.end());
        auto num = std::make_shared<numerator_oracle>();
        num->st = st_;
        numerator_handle_ = IOP.register_virtual_oracle(domain, numerator_degree, all, num);
    }
    oracle_handle numerator_handle() const { return numerator_handle_; }
    oracle_handle denominator_handle() const { return denominator_handle_; }
    void set_coefficients(const std::vector<FieldT> &coefficients)
    {
        if (coefficients.size() != st_->num_rationals) throw std::invalid_argument("Expected same number of random coefficients as oracles.");
        st_->coefficients = coefficients;
        st_->last_key.clear();
    }
    // :183-209 — the combined rational function itself (over the index domain)
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &numerator_evals, const std::vector<device_vector<FieldT>> &denominator_evals)
    {
        const auto nd = st_->pair(numerator_evals, denominator_evals);
        st_->last_key.clear();
        return dev::div<FieldT>(&nd.first, nd.second);
    }
};

template<typename FieldT>
class single_boundary_constraint : public virtual_oracle<FieldT> {                           // common/boundary_constraint.tcc: (f(x) - claimed) / (x - point)
    field_subset<FieldT> codeword_domain_;
    FieldT eval_point_, oracle_evaluation_;
public:
    explicit single_boundary_constraint(const field_subset<FieldT> &L) : codeword_domain_(L), eval_point_(field_host<FieldT>::zero()), oracle_evaluation_(field_host<FieldT>::zero()) {}
    void set_evaluation_point_and_eval(const FieldT &eval_point, const FieldT &oracle_eval) { eval_point_ = eval_point; oracle_evaluation_ = oracle_eval; }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        typedef field_host<FieldT> H;
        if (c.size() != 1) throw std::invalid_argument("Single Boundary Constraint: Expected exactly 1 constituent oracle.");
        if (H::element_in_domain(codeword_domain_, eval_point_)) throw std::logic_error("the evaluation point lies in the codeword domain");
        const device_vector<FieldT> numerator = dev::lincomb_affine<FieldT>(c, { H::neg(H::one()) }, oracle_evaluation_);      // claimed - f
        return dev::div<FieldT>(&numerator, dev::domain_offsets<FieldT>(D, eval_point_));                                      // / (point - x)
    }
};

template<typename FieldT>
class sumcheck_constraint_oracle : public virtual_oracle<FieldT> {                           // rational_sumcheck.tcc:9-137, constituents (p, N, D)
    field_subset<FieldT> summation_domain_, codeword_domain_;
    FieldT claimed_sum_;
public:
    sumcheck_constraint_oracle(const field_subset<FieldT> &K, const field_subset<FieldT> &L) : summation_domain_(K), codeword_domain_(L), claimed_sum_(field_host<FieldT>::zero()) {}
    void set_claimed_sum(const FieldT &claimed_sum) { claimed_sum_ = claimed_sum; }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    std::size_t smallest_window() const override { return summation_domain_.num_elements(); }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("sumcheck_constraint_oracle has three constituent oracles");
        const field_subset<FieldT> L = dist::local_domain(D);
        const field_subset<FieldT> &K = summation_domain_;
        device_vector<FieldT> out(L.num_elements());
        if (dev::additive(L)) {
            if (K.dimension() > L.dimension()) throw std::invalid_argument("the summation domain exceeds this rank's part of the codeword domain");
            for (std::size_t i = 0; i < K.dimension(); ++i)
                if (std::memcmp(&K.basis()[i], &L.basis()[i], sizeof(FieldT)) != 0) throw std::invalid_argument("the summation domain must be spanned by a prefix of the codeword domain's basis");
            const device_vector<FieldT> xinv = dev::div<FieldT>(nullptr, dev::domain_offsets<FieldT>(L, field_host<FieldT>::zero()));
            check(iopx_rational_sumcheck_constraint_gf192_dev(c[0].words(), c[1].words(), c[2].words(), xinv.words(), dev::basis_words(L), L.dimension(), dev::shift_words(L),
                                                              K.dimension(), dev::shift_words(K), detail::words(&claimed_sum_), out.words()));
        } else {
            check(iopx_rational_sumcheck_constraint_fp3_dev(c[0].words(), c[1].words(), c[2].words(), L.dimension(), dev::gen_words(L), dev::shift_words(L), K.dimension(),
                                                            dev::shift_words(K), detail::words(&claimed_sum_), out.words()));
        }
        return out;
    }
};

// ---- protocols ------------------------------------------------------------------------------------------------------------------
template<typename FieldT>
class rational_sumcheck_protocol {                                                            // rational_sumcheck.tcc:139-274
    bcs_prover<FieldT> &IOP_;
    domain_handle codeword_domain_handle_;
    field_subset<FieldT> K_, L_;
    std::size_t reextended_oracle_degree_, constraint_oracle_degree_;
    oracle_handle numerator_handle_, denominator_handle_, reextended_oracle_handle_, constraint_oracle_handle_;
    std::shared_ptr<sumcheck_constraint_oracle<FieldT>> constraint_oracle_;
    FieldT claimed_sum_;
public:
    rational_sumcheck_protocol(bcs_prover<FieldT> &IOP, const domain_handle &summation_domain_handle, const domain_handle &codeword_domain_handle,
                               std::size_t numerator_degree_bound, std::size_t denominator_degree_bound)
        : IOP_(IOP), codeword_domain_handle_(codeword_domain_handle), K_(IOP.get_domain(summation_domain_handle)), L_(IOP.get_domain(codeword_domain_handle)),
          claimed_sum_(field_host<FieldT>::zero())
    {
        reextended_oracle_degree_ = K_.num_elements() - 1;
        constraint_oracle_degree_ = std::max(numerator_degree_bound, denominator_degree_bound + K_.num_elements() - 1) - K_.num_elements();
    }
    void register_summation_oracle(const oracle_handle &numerator_handle, const oracle_handle &denominator_handle) { numerator_handle_ = numerator_handle; denominator_handle_ = denominator_handle; }
    void register_proof()
    {
        reextended_oracle_handle_ = IOP_.register_oracle("rational sumcheck reextension", codeword_domain_handle_, reextended_oracle_degree_, false);
        constraint_oracle_ = std::make_shared<sumcheck_constraint_oracle<FieldT>>(K_, L_);
        constraint_oracle_handle_ = IOP_.register_virtual_oracle(codeword_domain_handle_, constraint_oracle_degree_,
                                                                 { reextended_oracle_handle_, numerator_handle_, denominator_handle_ }, constraint_oracle_);
    }
    // :222-252 — interpolate over K, take the sum off the polynomial (its constant term times |K| / eps times its top coefficient),
    // re-extend the rest
    void calculate_and_submit_proof(const device_vector<FieldT> &rational_function_over_summation_domain)
    {
        typedef field_host<FieldT> H;
        const std::size_t n = K_.num_elements();
        const device_vector<FieldT> coeffs = dev::IFFT<FieldT>(rational_function_over_summation_domain, K_);
        device_vector<FieldT> rest;
        if (dev::additive(K_)) {
            const FieldT eps = H::vanishing_derivative(K_, H::zero());
            claimed_sum_ = H::mul(eps, coeffs.slice(n - 1, 1).to_host()[0]);
            rest = coeffs.slice(0, n - 1);
        } else {
            claimed_sum_ = H::mul(coeffs.slice(0, 1).to_host()[0], H::from_uint(n));
            rest = coeffs.slice(1, n - 1);
        }
        IOP_.submit_oracle(reextended_oracle_handle_, oracle<FieldT>(dev::FFT<FieldT>(rest, n - 1, L_)));
        constraint_oracle_->set_claimed_sum(claimed_sum_);
    }
    const FieldT &get_claimed_sum() const { return claimed_sum_; }
    std::vector<oracle_handle> get_all_oracle_handles() const { return { reextended_oracle_handle_, constraint_oracle_handle_ }; }
};

template<typename FieldT>
class holographic_multi_lincheck {                                                            // holographic_lincheck.tcc:113-580, non-zk
    typedef std::vector<device_vector<FieldT>> matrix_index;                                 // (row, col, val, row*col) over the index domain
    bcs_prover<FieldT> &IOP_;
    domain_handle codeword_domain_handle_, summation_domain_handle_, index_domain_handle_;
    const std::vector<sparse_matrix<FieldT>> *matrices_T_;
    std::size_t repetitions_, num_matrices_, lincheck_degree_;
    field_subset<FieldT> L_, H_, K_;
    std::vector<oracle_handle> constituent_oracle_handles_;
    std::vector<std::shared_ptr<batch_sumcheck_protocol<FieldT>>> sumcheck_H_;
    std::vector<std::shared_ptr<holographic_multi_lincheck_virtual_oracle<FieldT>>> lincheck_oracles_;
    std::vector<std::shared_ptr<single_boundary_constraint<FieldT>>> t_boundary_constraint_;
    const std::vector<matrix_index> *index_evals_over_K_ = nullptr;
    std::vector<std::vector<std::shared_ptr<single_matrix_denominator<FieldT>>>> matrix_denominators_;
    std::vector<std::vector<oracle_handle>> matrix_numerator_handles_, matrix_denominator_handles_;
    std::vector<std::shared_ptr<rational_sumcheck_protocol<FieldT>>> sumcheck_K_;
    std::vector<verifier_random_message_handle> alpha_handle_, random_coefficient_handle_, beta_handle_;
    std::vector<oracle_handle> t_oracle_handle_, t_boundary_constraint_handle_;
    std::vector<prover_message_handle> M_at_alpha_beta_;
    std::vector<std::shared_ptr<rational_linear_combination<FieldT>>> rational_linear_combination_;
    std::vector<std::vector<FieldT>> r_Mz_;
public:
    holographic_multi_lincheck(bcs_prover<FieldT> &IOP, const domain_handle &codeword_domain_handle, const domain_handle &summation_domain_handle,
                               const std::vector<sparse_matrix<FieldT>> *transposed_matrices, const oracle_handle &fz_handle, const std::vector<oracle_handle> &Mz_handles,
                               std::size_t repetitions)
        : IOP_(IOP), codeword_domain_handle_(codeword_domain_handle), summation_domain_handle_(summation_domain_handle), matrices_T_(transposed_matrices),
          repetitions_(repetitions), num_matrices_(transposed_matrices->size()), L_(IOP.get_domain(codeword_domain_handle)), H_(IOP.get_domain(summation_domain_handle))
    {
        if (num_matrices_ < 1) throw std::invalid_argument("multi_lincheck expects at least one matrix");
        if (Mz_handles.size() != num_matrices_) throw std::invalid_argument("inconsistent number of Mz_handles and matrices passed into multi lincheck.");
        constituent_oracle_handles_.push_back(fz_handle);
        constituent_oracle_handles_.insert(constituent_oracle_handles_.end(), Mz_handles.begin(), Mz_handles.end());
        lincheck_degree_ = H_.num_elements() + std::max(IOP.get_oracle_degree(fz_handle), IOP.get_oracle_degree(Mz_handles[0])) - 1;      // :146-150
        for (std::size_t r = 0; r < repetitions; ++r) {
            sumcheck_H_.push_back(std::make_shared<batch_sumcheck_protocol<FieldT>>(IOP, summation_domain_handle, codeword_domain_handle, lincheck_degree_));
            lincheck_oracles_.push_back(std::make_shared<holographic_multi_lincheck_virtual_oracle<FieldT>>(L_, H_, num_matrices_));
            t_boundary_constraint_.push_back(std::make_shared<single_boundary_constraint<FieldT>>(L_));
        }
    }
    // :190-254.  index_evals_over_K[i] = (row, col, val, row*col) of matrix i over the index domain, device resident: the reference
    // recomputes them inside calculate_response_beta (:447-458, "TODO: Also index evals over K"); here they are part of the prover's index.
    void set_index_oracles(const domain_handle &indexed_domain_handle, const std::vector<std::vector<oracle_handle>> &indexed_handles, const std::vector<matrix_index> *index_evals_over_K)
    {
        if (indexed_handles.size() != num_matrices_) throw std::invalid_argument("Incorrect number of sets of indexed oracles");
        for (auto &hs : indexed_handles) if (hs.size() != 4) throw std::invalid_argument("Incorrect number of indexed oracles within set");
        index_domain_handle_ = indexed_domain_handle;
        K_ = IOP_.get_domain(indexed_domain_handle);
        index_evals_over_K_ = index_evals_over_K;
        const std::size_t single = K_.num_elements();
        const std::size_t combined_numerator_degree = single + (num_matrices_ - 1) * single - (num_matrices_ - 1);
        const std::size_t combined_denominator_degree = num_matrices_ * single - (num_matrices_ - 1);
        for (std::size_t r = 0; r < repetitions_; ++r) {
            matrix_denominators_.emplace_back();
            matrix_numerator_handles_.emplace_back();
            matrix_denominator_handles_.emplace_back();
            for (std::size_t i = 0; i < num_matrices_; ++i) {
                auto d = std::make_shared<single_matrix_denominator<FieldT>>();
                matrix_denominators_[r].push_back(d);
                matrix_numerator_handles_[r].push_back(indexed_handles[i][2]);                // val
                // cached: the combined numerator and the combined denominator both read them (prover-side choice, not in the transcript)
                matrix_denominator_handles_[r].push_back(IOP_.register_virtual_oracle(codeword_domain_handle_, single,
                                                                                      { indexed_handles[i][0], indexed_handles[i][1], indexed_handles[i][3] }, d, true));
            }
            sumcheck_K_.push_back(std::make_shared<rational_sumcheck_protocol<FieldT>>(IOP_, indexed_domain_handle, codeword_domain_handle_, combined_numerator_degree,
                                                                                       combined_denominator_degree));
        }
    }
    void register_challenge_alpha()                                                          // :256-265
    {
        for (std::size_t r = 0; r < repetitions_; ++r) alpha_handle_.push_back(IOP_.register_verifier_random_message(1));
        for (std::size_t r = 0; r < repetitions_; ++r) random_coefficient_handle_.push_back(IOP_.register_verifier_random_message(num_matrices_));
    }
    void register_response_alpha()                                                           // :267-300
    {
        for (std::size_t r = 0; r < repetitions_; ++r) {
            const oracle_handle t = IOP_.register_oracle("lincheck_t", codeword_domain_handle_, H_.num_elements(), false);
            t_oracle_handle_.push_back(t);
            std::vector<oracle_handle> constituents = constituent_oracle_handles_;
            constituents.push_back(t);
            sumcheck_H_[r]->attach_oracle_for_summing(IOP_.register_virtual_oracle(codeword_domain_handle_, lincheck_degree_, constituents, lincheck_oracles_[r]));
        }
    }
    void register_challenge_beta()                                                           // :302-310
    {
        for (std::size_t r = 0; r < repetitions_; ++r) beta_handle_.push_back(IOP_.register_verifier_random_message(1));
        for (std::size_t r = 0; r < repetitions_; ++r) sumcheck_H_[r]->register_challenge();
    }
    void register_response_beta()                                                            // :312-366
    {
        for (std::size_t r = 0; r < repetitions_; ++r) M_at_alpha_beta_.push_back(IOP_.register_prover_message(1));
        for (std::size_t r = 0; r < repetitions_; ++r) {
            auto rlc = std::make_shared<rational_linear_combination<FieldT>>(IOP_, num_matrices_, matrix_numerator_handles_[r], matrix_denominator_handles_[r]);
            rational_linear_combination_.push_back(rlc);
            sumcheck_K_[r]->register_summation_oracle(rlc->numerator_handle(), rlc->denominator_handle());
            t_boundary_constraint_handle_.push_back(IOP_.register_virtual_oracle(codeword_domain_handle_, H_.num_elements() - 1, { t_oracle_handle_[r] }, t_boundary_constraint_[r]));
            sumcheck_H_[r]->register_proof();
            sumcheck_K_[r]->register_proof();
        }
    }
    void calculate_response_alpha()                                                          // :381-417
    {
        r_Mz_.assign(repetitions_, {});
        for (std::size_t r = 0; r < repetitions_; ++r) {
            const FieldT alpha = IOP_.obtain_verifier_random_message(alpha_handle_[r])[0];
            r_Mz_[r] = IOP_.obtain_verifier_random_message(random_coefficient_handle_[r]);
            const device_vector<FieldT> p_alpha_over_H = dev::lagrange_evals<FieldT>(alpha, H_, H_);                          // unnormalised (:393-397)
            const device_vector<FieldT> p_alpha_M_over_H(H_.num_elements());                                                   // compute_p_alpha_M (common.tcc:5-38)
            for (std::size_t m = 0; m < num_matrices_; ++m) (*matrices_T_)[m].times_vector(p_alpha_over_H, p_alpha_M_over_H, &r_Mz_[r][m], m > 0);
            IOP_.submit_oracle(t_oracle_handle_[r], oracle<FieldT>(dev::reextend_packed<FieldT>(p_alpha_M_over_H, 1, H_, L_)[0]));       // IFFT over H, FFT over L (:33, :410)
            lincheck_oracles_[r]->set_challenge(alpha, r_Mz_[r]);
        }
    }
    void calculate_response_beta()                                                           // :430-513
    {
        typedef field_host<FieldT> F;
        for (std::size_t r = 0; r < repetitions_; ++r) {
            const FieldT alpha = IOP_.obtain_verifier_random_message(alpha_handle_[r])[0], beta = IOP_.obtain_verifier_random_message(beta_handle_[r])[0];
            const FieldT shift = F::mul(F::vanishing_eval(H_, alpha), F::vanishing_eval(H_, beta));                            // :480-499
            std::vector<FieldT> coefficients;
            for (std::size_t i = 0; i < num_matrices_; ++i) coefficients.push_back(F::mul(shift, r_Mz_[r][i]));
            rational_linear_combination_[r]->set_coefficients(coefficients);
            for (auto &d : matrix_denominators_[r]) d->set_challenge(beta, alpha);                                             // :501-513
        }
        for (std::size_t r = 0; r < repetitions_; ++r) {
            const FieldT beta = IOP_.obtain_verifier_random_message(beta_handle_[r])[0];
            std::vector<device_vector<FieldT>> numerators, denominators;
            for (std::size_t i = 0; i < num_matrices_; ++i) {
                const matrix_index &ev = (*index_evals_over_K_)[i];
                numerators.push_back(ev[2]);
                denominators.push_back(matrix_denominators_[r][i]->evaluated_contents({ ev[0], ev[1], ev[3] }));
            }
            sumcheck_K_[r]->calculate_and_submit_proof(rational_linear_combination_[r]->evaluated_contents(numerators, denominators));
            const FieldT M_at_alpha_beta = sumcheck_K_[r]->get_claimed_sum();
            IOP_.submit_prover_message(M_at_alpha_beta_[r], { M_at_alpha_beta });
            t_boundary_constraint_[r]->set_evaluation_point_and_eval(beta, M_at_alpha_beta);
            sumcheck_H_[r]->calculate_and_submit_proof();
        }
    }
    std::vector<oracle_handle> get_all_oracle_handles() const                                // :550-580
    {
        std::vector<oracle_handle> out;
        for (std::size_t r = 0; r < repetitions_; ++r) {
            out.push_back(t_oracle_handle_[r]);
            out.push_back(t_boundary_constraint_handle_[r]);
            for (auto &h : sumcheck_H_[r]->get_all_oracle_handles()) out.push_back(h);
            for (auto &h : sumcheck_K_[r]->get_all_oracle_handles()) out.push_back(h);
        }
        return out;
    }
};

// matrix_indexer::compute_oracles_over_K (fractal_indexer.tcc:47-121) on the device: gathers of the matrix domain's elements by the
// entries' row / column, the values scaled by 1 / u_H(col, col) = 1 / (DZ_H)(col), padding, and the transposition swap.
// Returns [row, col, val, row*col] over the index domain K.
template<typename FieldT>
std::vector<device_vector<FieldT>> matrix_index_over_K(const sparse_matrix<FieldT> &M, const field_subset<FieldT> &K, const field_subset<FieldT> &H, std::size_t input_variable_dim)
{
    typedef field_host<FieldT> F;
    M.to_device();
    const std::size_t nnz = (std::size_t)M.row_ptr.back(), pad = K.num_elements() - nnz;
    std::vector<uint64_t> row_index(nnz), col_index(nnz);
    for (std::size_t r = 0; r < M.rows; ++r)
        for (uint64_t t = M.row_ptr[r]; t < M.row_ptr[r + 1]; ++t) { row_index[t] = r; col_index[t] = H.reindex_by_subset(input_variable_dim, M.col[t]); }
    const device_vector<FieldT> H_elements = dev::domain_elements<FieldT>(H);
    const device_vector<FieldT> row_evals = dev::gathered<FieldT>(H_elements, row_index), col_evals = dev::gathered<FieldT>(H_elements, col_index);
    const device_vector<FieldT> row_times_col = dev::mul<FieldT>(row_evals, col_evals);
    const device_vector<FieldT> coeff = M.d_coeff.slice(0, nnz);
    device_vector<FieldT> val_evals;
    if (dev::additive(H)) {                                   // (DZ_H) is the constant linear coefficient
        val_evals = dev::scaled<FieldT>(coeff, F::inverse(F::vanishing_derivative(H, F::zero())));
    } else {                                                  // (DZ_H)(c) = |H| c^(|H| - 1) = |H| shift^|H| / c on the coset
        const FieldT scale = F::inverse(F::mul(F::from_uint(H.num_elements()), F::pow(H.shift(), H.num_elements())));
        val_evals = dev::scaled<FieldT>(dev::mul<FieldT>(coeff, col_evals), scale);
    }
    const std::vector<FieldT> h0 = H_elements.slice(0, 1).to_host(), k0 = dev::domain_elements<FieldT>(K).slice(0, 1).to_host();
    auto padded = [&](const device_vector<FieldT> &t, const FieldT &fill) {
        device_vector<FieldT> out(K.num_elements());
        if (nnz) out.slice(0, nnz).copy_from(t);
        if (pad) out.slice(nnz, pad).copy_from(device_vector<FieldT>(device_array<FieldT>::from_host(std::vector<FieldT>(pad, fill))));
        return out;
    };
    const device_vector<FieldT> rows = padded(row_evals, h0[0]), cols = padded(col_evals, h0[0]);
    const device_vector<FieldT> vals = padded(val_evals, F::zero()), rcs = padded(row_times_col, F::mul(k0[0], k0[0]));
    return { cols, rows, vals, rcs };                          // "We are dealing with the transpose"
}

template<typename FieldT>
class matrix_indexer {                                                                        // fractal_indexer.tcc: the index of M' = M^T scaled by u_H(col, col)
    bcs_prover<FieldT> &IOP_;
    domain_handle codeword_domain_handle_;
    field_subset<FieldT> K_, H_, L_;
    std::size_t input_variable_dim_;
    const sparse_matrix<FieldT> &matrix_;
    std::vector<oracle_handle> handles_;
public:
    matrix_indexer(bcs_prover<FieldT> &IOP, const domain_handle &index_domain_handle, const domain_handle &matrix_domain_handle, const domain_handle &codeword_domain_handle,
                   std::size_t input_variable_dim, const sparse_matrix<FieldT> &matrix)
        : IOP_(IOP), codeword_domain_handle_(codeword_domain_handle), K_(IOP.get_domain(index_domain_handle)), H_(IOP.get_domain(matrix_domain_handle)),
          L_(IOP.get_domain(codeword_domain_handle)), input_variable_dim_(input_variable_dim), matrix_(matrix) {}
    std::vector<oracle_handle> register_oracles()                                            // :29-45: row, col, val, row*col
    {
        if (K_.num_elements() < (std::size_t)matrix_.row_ptr.back()) throw std::logic_error("index domain smaller than the number of non-zero entries");
        for (int i = 0; i < 4; ++i) handles_.push_back(IOP_.register_index_oracle(codeword_domain_handle_, K_.num_elements()));
        return handles_;
    }
    std::vector<device_vector<FieldT>> compute_oracles()                                     // :123-156
    {
        const std::vector<device_vector<FieldT>> over_K = matrix_index_over_K<FieldT>(matrix_, K_, H_, input_variable_dim_);
        const device_vector<FieldT> packed(4 * K_.num_elements());
        for (int i = 0; i < 4; ++i) packed.slice(i * K_.num_elements(), K_.num_elements()).copy_from(over_K[i]);
        const std::vector<device_vector<FieldT>> codewords = dev::reextend_packed<FieldT>(packed, 4, K_, L_);
        for (int i = 0; i < 4; ++i) IOP_.submit_oracle(handles_[i], oracle<FieldT>(codewords[i]));
        return over_K;
    }
};

template<typename FieldT>
class fractal_iop {                                                                           // fractal_hiop.tcc:218-346
    bcs_prover<FieldT> &IOP_;
    const fractal_snark_parameters<FieldT> &params_;
    domain_handle index_domain_handle_, matrix_domain_handle_, codeword_domain_handle_;
    field_subset<FieldT> quotient_map_domain_;
    std::vector<std::shared_ptr<matrix_indexer<FieldT>>> matrix_indexers_;
    std::vector<std::vector<oracle_handle>> indexed_handles_;
    std::shared_ptr<encoded_aurora_protocol<FieldT>> protocol_;
    std::shared_ptr<holographic_multi_lincheck<FieldT>> lincheck_;
    std::shared_ptr<LDT_instance_reducer<FieldT>> LDT_reducer_;
    std::vector<std::vector<device_vector<FieldT>>> rebuilt_index_evals_;
public:
    fractal_iop(bcs_prover<FieldT> &IOP, const r1cs_constraint_system<FieldT> &constraint_system, const fractal_snark_parameters<FieldT> &params,
                const std::vector<std::vector<device_vector<FieldT>>> *index_evals_over_K = nullptr)
        : IOP_(IOP), params_(params)
    {
        const field_subset<FieldT> index_domain((std::size_t)1 << params.index_domain_dim_), matrix_domain(params.num_constraints_);
        const FieldT codeword_domain_shift = field_subset<FieldT>((std::size_t)1 << params.codeword_domain_dim_).element_outside_of_subset();
        const field_subset<FieldT> codeword_domain = dist::mark_codeword_domain(field_subset<FieldT>((std::size_t)1 << params.codeword_domain_dim_, codeword_domain_shift),
                                                                                (std::size_t)1 << params.localization_parameters_[0]);
        index_domain_handle_ = IOP.register_domain(index_domain);
        matrix_domain_handle_ = IOP.register_domain(matrix_domain);
        codeword_domain_handle_ = IOP.register_domain(codeword_domain);
        quotient_map_domain_ = codeword_domain.get_subset_of_order((std::size_t)1 << params.localization_parameters_[0]);
        // register_index_oracles (:277-300); libff::log2(num_inputs)
        const std::size_t input_variable_dim = detail::log2_ceil(constraint_system.num_inputs());
        const sparse_matrix<FieldT> *M[3] = { &constraint_system.A, &constraint_system.B, &constraint_system.C };
        for (int q = 0; q < 3; ++q) {
            matrix_indexers_.push_back(std::make_shared<matrix_indexer<FieldT>>(IOP, index_domain_handle_, matrix_domain_handle_, codeword_domain_handle_, input_variable_dim, *M[q]));
            indexed_handles_.push_back(matrix_indexers_.back()->register_oracles());
        }
        IOP.set_round_parameters(quotient_map_domain_);
        IOP.signal_index_registrations_done();
        // :253-275
        protocol_ = std::make_shared<encoded_aurora_protocol<FieldT>>(IOP, matrix_domain_handle_, matrix_domain_handle_, codeword_domain_handle_, constraint_system, 0, true);
        lincheck_ = std::make_shared<holographic_multi_lincheck<FieldT>>(IOP, codeword_domain_handle_, matrix_domain_handle_, protocol_->transposed_matrices(), protocol_->fz_handle(),
                                                                         protocol_->Mz_handles(), params.holographic_lincheck_repetitions_);
        if (index_evals_over_K && input_variable_dim != protocol_->input_variable_domain().dimension()) {
            // Reference quirk F15: the indexer reindexes columns with libff::log2(num_inputs) (fractal_hiop.tcc:279) while the lincheck rebuilds
            // the index evaluations with log2(num_inputs + 1) (holographic_lincheck.tcc:447-458 via r1cs_rs_iop.tcc:352); they differ for
            // num_inputs = 1 only (multiplicative domains), where the reference's own proof is rejected.  Follow it.
            for (int q = 0; q < 3; ++q) rebuilt_index_evals_.push_back(matrix_index_over_K<FieldT>(*M[q], index_domain, matrix_domain, protocol_->input_variable_domain().dimension()));
            index_evals_over_K = &rebuilt_index_evals_;
        }
        lincheck_->set_index_oracles(index_domain_handle_, indexed_handles_, index_evals_over_K);
        LDT_reducer_ = std::make_shared<LDT_instance_reducer<FieldT>>(IOP, codeword_domain_handle_, params.num_output_LDT_instances_, params.max_LDT_tested_degree_bound_);
        IOP.set_round_parameters(quotient_map_domain_);
    }
    void register_interactions()                                                             // :302-325
    {
        lincheck_->register_challenge_alpha();
        IOP_.set_round_parameters(quotient_map_domain_);
        lincheck_->register_response_alpha();
        lincheck_->register_challenge_beta();
        lincheck_->register_response_beta();
        IOP_.set_round_parameters(quotient_map_domain_);
        std::vector<oracle_handle> handles = lincheck_->get_all_oracle_handles();
        for (auto &h : protocol_->witness_and_rowcheck_handles()) handles.push_back(h);      // r1cs_rs_iop.tcc:650-668
        LDT_reducer_->register_interactions(handles, params_.localization_parameters_, params_.fri_interactive_repetitions_, params_.fri_query_repetitions_);
    }
    void register_queries() { LDT_reducer_->register_queries(); }
    std::vector<std::vector<device_vector<FieldT>>> produce_index()                          // :306-314
    {
        std::vector<std::vector<device_vector<FieldT>>> over_K;
        for (auto &mi : matrix_indexers_) over_K.push_back(mi->compute_oracles());
        IOP_.signal_index_submissions_done();
        return over_K;
    }
    // produce_proof in two halves (as aurora_iop's): the index and the witness oracles' kernels first, the query registrations while the GPU works
    bool witness_submitted_ = false;
    void submit_witness(const std::vector<FieldT> &primary_input, const std::vector<FieldT> &auxiliary_input, const bcs_prover_index<FieldT> &index,
                        const device_vector<FieldT> *d_assignment = nullptr)
    {
        IOP_.submit_prover_index(index);
        protocol_->submit_witness_oracles(primary_input, auxiliary_input, d_assignment);
        witness_submitted_ = true;
    }
    void produce_proof(const std::vector<FieldT> &primary_input, const std::vector<FieldT> &auxiliary_input, const bcs_prover_index<FieldT> &index,
                       const device_vector<FieldT> *d_assignment = nullptr)                  // :316-329, r1cs_rs_iop.tcc:618-627
    {
        if (!witness_submitted_) submit_witness(primary_input, auxiliary_input, index, d_assignment);
        IOP_.signal_prover_round_done();
        lincheck_->calculate_response_alpha();
        IOP_.signal_prover_round_done();
        lincheck_->calculate_response_beta();
        IOP_.signal_prover_round_done();
        LDT_reducer_->calculate_and_submit_proof();
    }
};

// fractal_snark_indexer (fractal_snark.tcc:114-133): (prover index, verifier index) — the twelve index oracles over the codeword domain
// with their Merkle tree (and their evaluations over the index domain) in HBM / the tree's root
template<typename FieldT>
std::pair<bcs_prover_index<FieldT>, bcs_verifier_index> fractal_snark_indexer(const r1cs_constraint_system<FieldT> &constraint_system,
                                                                              const fractal_snark_parameters<FieldT> &parameters)
{
    constraint_system.prepare_device();
    bcs_prover<FieldT> IOP(parameters.pow_bits_);
    fractal_iop<FieldT> full_protocol(IOP, constraint_system, parameters);
    IOP.seal_interaction_registrations();
    IOP.seal_query_registrations();
    const auto over_K = full_protocol.produce_index();
    bcs_prover_index<FieldT> index = IOP.get_prover_index();
    index.index_evals_over_K = over_K;
    return { index, IOP.get_verifier_index() };
}

// fractal_snark_prover (fractal_snark.tcc:135-162): the transcript without the index's roots
template<typename FieldT>
bcs_transformation_transcript<FieldT> fractal_snark_prover(const bcs_prover_index<FieldT> &index, const r1cs_constraint_system<FieldT> &constraint_system,
                                                           const r1cs_primary_input<FieldT> &primary_input, const r1cs_auxiliary_input<FieldT> &auxiliary_input,
                                                           const fractal_snark_parameters<FieldT> &parameters, const device_vector<FieldT> *d_assignment = nullptr)
{
    constraint_system.prepare_device();
    bcs_prover<FieldT> IOP(parameters.pow_bits_, &index);
    fractal_iop<FieldT> full_protocol(IOP, constraint_system, parameters, &index.index_evals_over_K);
    full_protocol.register_interactions();
    IOP.seal_interaction_registrations();
    full_protocol.register_queries();
    IOP.seal_query_registrations();
    full_protocol.produce_proof(primary_input, auxiliary_input, index, d_assignment);
    return IOP.get_transcript();
}

// ... returning fractal_snark_prover(...).serialize() without building the structured transcript (bcs_prover::get_transcript_bytes)
template<typename FieldT>
std::string fractal_snark_prover_serialized(const bcs_prover_index<FieldT> &index, const r1cs_constraint_system<FieldT> &constraint_system,
                                            const r1cs_primary_input<FieldT> &primary_input, const r1cs_auxiliary_input<FieldT> &auxiliary_input,
                                            const fractal_snark_parameters<FieldT> &parameters, const device_vector<FieldT> *d_assignment = nullptr)
{
    constraint_system.prepare_device();
    bcs_prover<FieldT> IOP(parameters.pow_bits_, &index);
    fractal_iop<FieldT> full_protocol(IOP, constraint_system, parameters, &index.index_evals_over_K);
    full_protocol.register_interactions();
    IOP.seal_interaction_registrations();
    full_protocol.submit_witness(primary_input, auxiliary_input, index, d_assignment);      // round 1's kernels are in flight while the queries are registered
    full_protocol.register_queries();
    IOP.seal_query_registrations();
    full_protocol.produce_proof(primary_input, auxiliary_input, index, d_assignment);
    return IOP.get_transcript_bytes();
}

} // namespace libiop_amd
