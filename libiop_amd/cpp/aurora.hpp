// Aurora SNARK prover (non-zk, BLAKE2b) for C++ callers: the reference's composition, every vector in HBM.
//
//   aurora_snark_parameters / aurora_snark_prover        libiop/snark/aurora_snark.tcc:38-146
//   aurora_iop_parameters, aurora_iop                    libiop/protocols/aurora_iop.tcc:3-186, 262-344
//   encoded_aurora_protocol, fz_virtual_oracle           libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.tcc
//   multi_lincheck, multi_lincheck_virtual_oracle        libiop/protocols/encoded/lincheck/basic_lincheck{,_aux}.tcc
//   batch_sumcheck_protocol, sumcheck_g_oracle           libiop/protocols/encoded/sumcheck/sumcheck.tcc
//   rowcheck_ABC_virtual_oracle,
//   random_linear_combination_oracle                     libiop/protocols/encoded/common/{rowcheck,random_linear_combination}.tcc
//   LDT_instance_reducer, combined_LDT_virtual_oracle    libiop/protocols/ldt/ldt_reducer{,_aux}.tcc
//   FRI_protocol                                         libiop/protocols/ldt/fri/fri_ldt.tcc:260-548
//
// Registration order — which fixes rounds, Merkle trees and the hashchain's squeeze order — is the reference's.  The bodies call
// the device operators of include/libiop_amd.h; nothing codeword-sized is ever in a std::vector.  Zero knowledge is out of scope
// (masks and salts come from libsodium randomness: not reproducible).  Citations are relative to the reference tree.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "r1cs.hpp"

namespace libiop_amd {

// ---- device operators, dispatched on the domain type (FFT_over_field_subset and friends dispatch the same way, fft.tcc:407-475) ----
namespace dev {

template<typename FieldT> inline const uint64_t *basis_words(const field_subset<FieldT> &D) { return detail::words(D.basis().data()); }
// shift() and generator() return BY VALUE (field_subset.hpp:65-67): the holder keeps the copy alive for the full expression of the call
// it is an argument of.  Never store the converted pointer.
template<typename FieldT> struct held_words {
    FieldT v;
    operator const uint64_t *() const { return detail::words(&v); }
};
template<typename FieldT> inline held_words<FieldT> shift_words(const field_subset<FieldT> &D) { return held_words<FieldT>{ D.shift() }; }
template<typename FieldT> inline held_words<FieldT> gen_words(const field_subset<FieldT> &D) { return held_words<FieldT>{ D.generator() }; }
template<typename FieldT> inline bool additive(const field_subset<FieldT> &D) { return D.type() == affine_subspace_type; }

template<typename FieldT>
device_vector<FieldT> FFT(const device_vector<FieldT> &coeffs, std::size_t n_coeffs, const field_subset<FieldT> &D)            // fft.tcc:407-419
{
    if (D.distributed()) {                                                                   // dist.hpp: this rank's part of the codeword
        device_vector<FieldT> out(dist::local_size(D));
        if (additive(D)) {                                                                   // its coset range of the transform; phase 1 on the coefficients is replicated
            const auto range = dist::coset_range(D, detail::log2_ceil(n_coeffs));
            check(iopx_add_lde_gf192_dev(coeffs.words(), n_coeffs, basis_words(D), D.dimension(), shift_words(D), range.first, range.second, out.words()));
        } else {                                                                             // one ordinary transform over the rank's sub-coset
            if (n_coeffs > out.size()) throw std::invalid_argument("more coefficients than a residue class holds");
            const field_subset<FieldT> loc = dist::local_domain(D);
            check(iopx_mul_fft_fp3_dev(coeffs.words(), n_coeffs, loc.dimension(), gen_words(loc), shift_words(loc), out.words()));
        }
        return out;
    }
    device_vector<FieldT> out(D.num_elements());
    if (additive(D)) { check(iopx_add_fft_gf192_dev(coeffs.words(), n_coeffs, basis_words(D), D.dimension(), shift_words(D), out.words())); return out; }
    // the windows the prover will ask for (dist::window_collector) leave the transform's last pass together with the codeword
    dist::window_collector<FieldT> *col = dist::active_collector<FieldT>();
    if (col && col->domain_elements == D.num_elements() && !col->wanted.empty()) {
        typename dist::window_collector<FieldT>::entry e;
        std::size_t firsts[2], log_strides[2];
        uint64_t *ptrs[2];
        std::size_t nw = 0;
        for (const dist::window &w : col->wanted) {
            if (nw == 2) break;
            e.windows.emplace_back(w, device_vector<FieldT>(w.count));
            firsts[nw] = w.first; log_strides[nw] = detail::log2_ceil(w.stride); ptrs[nw] = e.windows.back().second.words();
            ++nw;
        }
        check(iopx_mul_fft_fp3_windows_dev(coeffs.words(), n_coeffs, D.dimension(), gen_words(D), shift_words(D), out.words(), nw, firsts, log_strides, ptrs));
        e.codeword = out;
        col->produced.push_back(std::move(e));
        return out;
    }
    check(iopx_mul_fft_fp3_dev(coeffs.words(), n_coeffs, D.dimension(), gen_words(D), shift_words(D), out.words()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> IFFT(const device_vector<FieldT> &evals, const field_subset<FieldT> &D)                                   // fft.tcc:421-433
{
    if (D.distributed()) throw std::logic_error("inverse transform of a distributed vector: gather it first");
    if (evals.size() != D.num_elements()) throw std::invalid_argument("IFFT: evaluation count != domain size");
    device_vector<FieldT> out(D.num_elements());
    if (additive(D)) check(iopx_add_ifft_gf192_dev(evals.words(), basis_words(D), D.dimension(), shift_words(D), out.words()));
    else check(iopx_mul_ifft_fp3_dev(evals.words(), D.dimension(), gen_words(D), shift_words(D), out.words()));
    return out;
}

// FFT_over_field_subset(IFFT_over_field_subset(v, H), L) for `batch` vectors stored back to back.  When H is spanned by the first basis
// vectors of L the coefficient form is skipped (iopx_add_reextend_gf192_batch_dev); otherwise the two transforms run one after the other.
// H_is_first_coset: H is the first coset of its span inside L (same shift), so the first |H| entries of each codeword are the input itself.
template<typename FieldT>
std::vector<device_vector<FieldT>> reextend_packed(const device_vector<FieldT> &packed, std::size_t batch, const field_subset<FieldT> &H,
                                                   const field_subset<FieldT> &L, bool H_is_first_coset = false)
{
    const std::size_t n = H.num_elements();
    std::vector<device_vector<FieldT>> outs;
    bool prefix = additive(L) && H.dimension() <= L.dimension();
    if (prefix) for (std::size_t i = 0; i < H.dimension(); ++i) prefix = prefix && std::memcmp(&H.basis()[i], &L.basis()[i], sizeof(FieldT)) == 0;
    if (prefix) {
        std::vector<uint64_t *> ptrs;
        auto range = dist::coset_range(L, H.dimension());
        for (std::size_t k = 0; k < batch; ++k) outs.emplace_back(range.second << H.dimension());
        const bool copy_first = H_is_first_coset && range.first == 0 && range.second > 1 && H.shift() == L.shift();
        for (std::size_t k = 0; k < batch; ++k) {
            if (copy_first) outs[k].slice(0, n).copy_from(packed.slice(k * n, n));
            ptrs.push_back(outs[k].words() + (copy_first ? 3 * n : 0));
        }
        if (copy_first) { range.first = 1; range.second -= 1; }
        check(iopx_add_reextend_gf192_batch_dev(packed.words(), batch, basis_words(L), L.dimension(), H.dimension(), shift_words(H), shift_words(L), range.first,
                                                range.second, ptrs.data()));
        return outs;
    }
    for (std::size_t k = 0; k < batch; ++k) outs.push_back(FFT<FieldT>(IFFT<FieldT>(packed.slice(k * n, n), H), n, L));
    return outs;
}

// reextend_packed(a, batch_a, Ha, L) followed by reextend_packed(b, batch_b, Hb, L) in one call, when Ha and Hb are cosets of the same span of L's first
// basis vectors (iopx_add_reextend2_gf192_batch_dev: the forward passes and the shared last pass run over all the vectors); the separate calls otherwise.
template<typename FieldT>
std::vector<device_vector<FieldT>> reextend2_packed(const device_vector<FieldT> &a, std::size_t batch_a, const field_subset<FieldT> &Ha, const device_vector<FieldT> &b,
                                                    std::size_t batch_b, const field_subset<FieldT> &Hb, const field_subset<FieldT> &L)
{
    std::vector<device_vector<FieldT>> outs;
    bool together = additive(L) && additive(Ha) && additive(Hb) && Ha.dimension() == Hb.dimension() && Ha.dimension() > 0 && Ha.dimension() <= L.dimension();
    for (std::size_t i = 0; together && i < Ha.dimension(); ++i)
        together = std::memcmp(&Ha.basis()[i], &L.basis()[i], sizeof(FieldT)) == 0 && std::memcmp(&Hb.basis()[i], &L.basis()[i], sizeof(FieldT)) == 0;
    if (!together) {
        outs = reextend_packed<FieldT>(a, batch_a, Ha, L);
        for (auto &cw : reextend_packed<FieldT>(b, batch_b, Hb, L)) outs.push_back(cw);
        return outs;
    }
    const auto range = dist::coset_range(L, Ha.dimension());
    std::vector<uint64_t *> ptrs;
    for (std::size_t k = 0; k < batch_a + batch_b; ++k) { outs.emplace_back(range.second << Ha.dimension()); ptrs.push_back(outs.back().words()); }
    check(iopx_add_reextend2_gf192_batch_dev(a.words(), batch_a, shift_words(Ha), b.words(), batch_b, shift_words(Hb), basis_words(L), L.dimension(), Ha.dimension(),
                                             shift_words(L), range.first, range.second, ptrs.data()));
    return outs;
}

// { FFT_over_field_subset(poly, L) } followed by reextend_packed(packed, batch, H, L): one call when H is spanned by the first basis vectors
// of L and the polynomial has at most |H| coefficients (iopx_add_reextend_lde_gf192_batch_dev: the transforms share their last passes),
// the separate calls otherwise.  Same field elements either way.
template<typename FieldT>
std::vector<device_vector<FieldT>> FFT_and_reextend_packed(const device_vector<FieldT> &poly, const device_vector<FieldT> &packed, std::size_t batch,
                                                           const field_subset<FieldT> &H, const field_subset<FieldT> &L)
{
    bool prefix = additive(L) && additive(H) && H.dimension() <= L.dimension() && poly.size() <= H.num_elements() && H.dimension() > 0;
    if (prefix) for (std::size_t i = 0; i < H.dimension(); ++i) prefix = prefix && std::memcmp(&H.basis()[i], &L.basis()[i], sizeof(FieldT)) == 0;
    std::vector<device_vector<FieldT>> outs;
    if (!prefix) {
        outs.push_back(FFT<FieldT>(poly, poly.size(), L));
        for (auto &cw : reextend_packed<FieldT>(packed, batch, H, L)) outs.push_back(cw);
        return outs;
    }
    std::vector<uint64_t *> ptrs(batch + 1);
    const auto range = dist::coset_range(L, H.dimension());
    for (std::size_t k = 0; k <= batch; ++k) outs.emplace_back(range.second << H.dimension());
    for (std::size_t k = 0; k < batch; ++k) ptrs[k] = outs[1 + k].words();
    ptrs[batch] = outs[0].words();
    const uint64_t *coeffs[1] = { poly.words() };
    check(iopx_add_reextend_lde_gf192_batch_dev(packed.words(), batch, coeffs, poly.size(), 1, basis_words(L), L.dimension(), H.dimension(), shift_words(H),
                                                shift_words(L), range.first, range.second, ptrs.data()));
    return outs;
}

template<typename FieldT> device_vector<FieldT> IFFT_of_known_degree(const device_vector<FieldT> &evals, std::size_t degree, const field_subset<FieldT> &D);
template<typename FieldT> device_vector<FieldT> poly_div_vanishing(const device_vector<FieldT> &poly, std::size_t n_coeffs, const field_subset<FieldT> &D);

// polynomial_over_vanishing_polynomial(IFFT_of_known_degree(evals, degree, L), Z_H).first — the sumcheck's h (sumcheck.tcc:351-365).  Over a
// distributed L whose needed evaluations all sit on rank 0, rank 0 interpolates AND divides, and h's coefficients are broadcast
// (degree - |H| elements: the one codeword-derived exchange of a proof) instead of the interpolant's.
template<typename FieldT>
device_vector<FieldT> interpolate_and_divide(const device_vector<FieldT> &evals, std::size_t degree, const field_subset<FieldT> &L, const field_subset<FieldT> &H)
{
    const std::size_t k = detail::log2_ceil(degree);
    const bool on_rank0 = L.distributed() && degree > H.num_elements() &&
                          (additive(L) ? ((std::size_t)1 << k) <= dist::local_size(L) : (L.num_elements() >> k) % dist::ctx().world == 0);
    if (!on_rank0) return poly_div_vanishing<FieldT>(IFFT_of_known_degree<FieldT>(evals, degree, L), degree, H);
    const dist::context &c = dist::ctx();
    device_vector<FieldT> h(degree - H.num_elements());
    if (c.rank == 0) {
        const dist::one_rank_section alone;
        field_subset<FieldT> whole = L;
        whole.set_distributed(false);
        const device_vector<FieldT> poly = additive(L) ? IFFT<FieldT>(evals.slice(0, (std::size_t)1 << k), whole.get_subset_of_order((std::size_t)1 << k))
                                                       : IFFT_of_known_degree<FieldT>(evals, degree, dist::local_domain(L, 0));
        h = poly_div_vanishing<FieldT>(poly, degree, H);
    }
    dist::broadcast<FieldT>(h, 0);
    return h;
}

// IFFT_of_known_degree_over_field_subset (fft.tcc:435-475): 2^ceil(log2 degree) coefficients
template<typename FieldT>
device_vector<FieldT> IFFT_of_known_degree(const device_vector<FieldT> &evals, std::size_t degree, const field_subset<FieldT> &D)
{
    const std::size_t k = detail::log2_ceil(degree), count = (std::size_t)1 << k;
    if (D.distributed()) {
        // Subspaces read the first 2^k evaluations (fft.tcc:458-475): rank 0's head.  Cosets read every (|D| / 2^k)-th (fft.tcc:435-456):
        // all on rank 0 while that stride is a multiple of N, at stride / N in its sub-coset (whose shift is the domain's).  Rank 0
        // interpolates, the coefficients are broadcast; otherwise the vector is gathered first.
        const dist::context &c = dist::ctx();
        field_subset<FieldT> whole = D;
        whole.set_distributed(false);
        const bool on_rank0 = additive(D) ? count <= dist::local_size(D) : (D.num_elements() >> k) % c.world == 0;
        if (!on_rank0) return IFFT_of_known_degree<FieldT>(dist::gather<FieldT>(evals, D), degree, whole);
        device_vector<FieldT> out(count);
        if (c.rank == 0) {
            const dist::one_rank_section alone;
            if (additive(D)) out = IFFT<FieldT>(evals.slice(0, count), whole.get_subset_of_order(count));
            else out = IFFT_of_known_degree<FieldT>(evals, degree, dist::local_domain(D, 0));
        }
        dist::broadcast<FieldT>(out, 0);
        return out;
    }
    if (additive(D)) return IFFT<FieldT>(evals.slice(0, count), D.get_subset_of_order(count));
    device_vector<FieldT> out(count);
    check(iopx_mul_ifft_known_degree_fp3_dev(evals.words(), degree, D.dimension(), gen_words(D), shift_words(D), out.words()));
    return out;
}

// Over a distributed domain the rank's cosets are whole (dist.hpp), so its part folds as the local sub-domain with the same x_i and is the
// rank's part of f_(i+1); when the next domain is kept whole the parts are gathered.
template<typename FieldT>
device_vector<FieldT> fold(const device_vector<FieldT> &f, const field_subset<FieldT> &D_in, std::size_t coset_size, const FieldT &x_i, bool next_distributed = false)   // fri_aux.tcc:5-34
{
    if (D_in.distributed()) {
        const device_vector<FieldT> part = fold<FieldT>(f, dist::local_domain(D_in), coset_size, x_i);
        return next_distributed ? part : dist::gather_layout<FieldT>(part, D_in.type());
    }
    const field_subset<FieldT> &D = D_in;
    device_vector<FieldT> out(D.num_elements() / coset_size);
    if (additive(D)) check(iopx_fri_fold_add_gf192_dev(f.words(), basis_words(D), D.dimension(), shift_words(D), coset_size, detail::words(&x_i), out.words()));
    else check(iopx_fri_fold_mul_fp3_dev(f.words(), D.dimension(), gen_words(D), shift_words(D), coset_size, detail::words(&x_i), out.words()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> sub(const device_vector<FieldT> &a, const device_vector<FieldT> &b)
{
    if (a.size() != b.size()) throw std::invalid_argument("sub: size mismatch");
    device_vector<FieldT> out(a.size());
    if (field_host<FieldT>::additive()) check(iopx_gf192_add_dev(a.words(), b.words(), out.words(), a.size()));
    else check(iopx_fp3_sub_dev(a.words(), b.words(), out.words(), a.size()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> pow_table(std::size_t count, const FieldT &base, const FieldT &init)                                      // out[l] = init * base^l
{
    device_vector<FieldT> out(count);
    if (field_host<FieldT>::additive()) check(iopx_gf192_pow_table_dev(out.words(), count, detail::words(&base), detail::words(&init)));
    else check(iopx_fp3_pow_table_dev(out.words(), count, detail::words(&base), detail::words(&init)));
    return out;
}

// polynomial_over_vanishing_polynomial(P, Z_D).first: n_coeffs - |D| coefficients
template<typename FieldT>
device_vector<FieldT> poly_div_vanishing(const device_vector<FieldT> &poly, std::size_t n_coeffs, const field_subset<FieldT> &D)
{
    device_vector<FieldT> out(n_coeffs > D.num_elements() ? n_coeffs - D.num_elements() : 0);
    if (n_coeffs > D.num_elements()) {
        if (additive(D)) check(iopx_poly_div_vanishing_gf192_dev(poly.words(), n_coeffs, basis_words(D), D.dimension(), shift_words(D), out.words()));
        else check(iopx_poly_div_vanishing_fp3_dev(poly.words(), n_coeffs, D.dimension(), shift_words(D), out.words()));
    }
    return out;
}

// Virtual oracles whose POLYNOMIAL is wanted (the sumcheck's combined f, FRI's first oracle) are evaluated over the head of the codeword
// domain only (dist::head_domain: as many points as the polynomial has coefficients) instead of over all of it, when that is at least 4x
// fewer points and rank 0 holds them.  IOPX_HEAD_EVAL=0: the reference's schedule (every virtual oracle over the whole domain).
template<typename FieldT>
device_vector<FieldT> div(const device_vector<FieldT> *num, const device_vector<FieldT> &den)        // batch_inverse(_and_mul), utils.tcc:57-118
{
    device_vector<FieldT> out(den.size());
    auto fn = field_host<FieldT>::additive() ? iopx_gf192_div_dev : iopx_fp3_div_dev;
    check(fn(num ? num->words() : nullptr, den.words(), out.words(), den.size()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> vanishing_evals(const field_subset<FieldT> &S, const field_subset<FieldT> &D_in, const FieldT &constant)  // constant - Z_S(x) over D (this rank's part)
{
    const field_subset<FieldT> D = dist::local_domain(D_in);
    device_vector<FieldT> out(D.num_elements());
    if (additive(D))
        check(iopx_vanishing_evals_gf192_dev(basis_words(D), D.dimension(), shift_words(D), basis_words(S), S.dimension(), shift_words(S), detail::words(&constant),
                                             out.words()));
    else
        check(iopx_vanishing_evals_fp3_dev(D.dimension(), gen_words(D), shift_words(D), S.dimension(), shift_words(S), detail::words(&constant), out.words()));
    return out;
}

// is H (a subspace) spanned by the first basis vectors of L?
template<typename FieldT>
bool spanned_by_prefix(const field_subset<FieldT> &H, const field_subset<FieldT> &L)
{
    if (!additive(H) || !additive(L) || H.dimension() > L.dimension()) return false;
    for (std::size_t i = 0; i < H.dimension(); ++i) if (std::memcmp(&H.basis()[i], &L.basis()[i], sizeof(FieldT)) != 0) return false;
    return true;
}

inline bool head_evaluation_enabled()
{
    return iopx_get_option("IOPX_HEAD_EVAL", 1) != 0;   // looked up per proof (iopx_set_option): tests and bench.py prove the same instance both ways
}
template<typename FieldT>
bool use_head(const field_subset<FieldT> &D, std::size_t count)
{
    return head_evaluation_enabled() && count * 4 <= D.num_elements() && D.num_elements() % count == 0 && dist::head_on_rank0(D, count);
}

template<typename FieldT>
std::vector<const void *> pointers(const std::vector<device_vector<FieldT>> &v, std::size_t first = 0)
{
    std::vector<const void *> p;
    for (std::size_t i = first; i < v.size(); ++i) p.push_back(v[i].data());
    return p;
}

} // namespace dev

// ---- parameters (aurora_snark.tcc:38-101, aurora_iop.tcc:3-186, common_bcs_parameters.tcc:9-27; non-zk, heuristic FRI soundness,
// optimistic-heuristic LDT-reducer soundness: the settings of profiling/instrument_aurora_snark.cpp:209-217) ----
inline std::vector<std::size_t> localization_parameter_to_array(std::size_t localization_parameter, std::size_t codeword_domain_dim, std::size_t RS_extra_dimensions)
{
    const std::size_t num_reductions = ((codeword_domain_dim - RS_extra_dimensions - 1) / localization_parameter) + 1;      // fri_ldt.tcc:132-146
    std::vector<std::size_t> out(1, 1);
    for (std::size_t i = 1; i < num_reductions; ++i) out.push_back(localization_parameter);
    return out;
}

template<typename FieldT>
struct aurora_snark_parameters {
    std::size_t security_parameter_, RS_extra_dimensions_, num_constraints_, num_variables_, num_inputs_;
    std::size_t constraint_domain_dim_, variable_domain_dim_, summation_domain_dim_, codeword_domain_dim_;
    std::size_t pow_bits_, query_soundness_error_bits_, interactive_soundness_error_bits_;
    std::vector<std::size_t> localization_parameters_;
    std::size_t max_tested_degree_bound_, max_constraint_degree_bound_, multi_lincheck_repetitions_, absolute_proximity_parameter_;
    std::size_t num_output_LDT_instances_, fri_query_repetitions_, fri_interactive_repetitions_;

    static bool is_pow2(std::size_t n) { return n > 0 && (n & (n - 1)) == 0; }
    static std::size_t repetitions(double bits, double per) { const double r = std::ceil(-bits / per); return r < 1 ? 1 : (std::size_t)r; }

    aurora_snark_parameters(std::size_t num_constraints, std::size_t num_variables, std::size_t num_inputs, std::size_t security_parameter = 128,
                            std::size_t RS_extra_dimensions = 5, std::size_t FRI_localization_parameter = 2)
        : security_parameter_(security_parameter), RS_extra_dimensions_(RS_extra_dimensions), num_constraints_(num_constraints), num_variables_(num_variables),
          num_inputs_(num_inputs)
    {
        // bcs_prover (iop.hpp) hashes with 32-byte BLAKE2b digests = the reference's digest_len_bytes = 2 * security_parameter / 8
        // (blake2b.tcc:15) for 128 only; other values would prove with digests the reference does not use
        if (security_parameter != 128) throw std::invalid_argument("libiop_amd: security_parameter must be 128 (32-byte BLAKE2b digests)");
        if (!is_pow2(num_constraints)) throw std::invalid_argument("number of constraints in the constraint system must a power of two.");
        if (!is_pow2(num_variables + 1)) throw std::invalid_argument("number of variables in the constraint system must be one less than a power of two.");
        if (!is_pow2(num_inputs + 1)) throw std::invalid_argument("number of inputs in the constraint system must be one less than a power of two.");
        constraint_domain_dim_ = detail::log2_ceil(num_constraints);                          // aurora_iop.tcc:35-36
        variable_domain_dim_ = detail::log2_ceil(num_variables + 1);
        summation_domain_dim_ = std::max(constraint_domain_dim_, variable_domain_dim_);
        codeword_domain_dim_ = summation_domain_dim_ + RS_extra_dimensions;                   // :39-43 (make_zk false)
        pow_bits_ = constraint_domain_dim_ + 3;                                               // common_bcs_parameters.tcc:23-25
        query_soundness_error_bits_ = security_parameter + 1 - pow_bits_;                     // aurora_iop.tcc:77
        interactive_soundness_error_bits_ = security_parameter + 3;                           // :78
        localization_parameters_ = localization_parameter_to_array(FRI_localization_parameter, codeword_domain_dim_, RS_extra_dimensions);
        max_tested_degree_bound_ = (std::size_t)1 << summation_domain_dim_;                   // r1cs_rs_iop.tcc:56-63
        max_constraint_degree_bound_ = std::max(2 * ((std::size_t)1 << summation_domain_dim_) - 1, 2 * ((std::size_t)1 << constraint_domain_dim_) - 1);
        const double fbits = (double)field_host<FieldT>::soundness_bits();
        multi_lincheck_repetitions_ = repetitions((double)interactive_soundness_error_bits_, (double)constraint_domain_dim_ - fbits);     // basic_lincheck.tcc:52-56
        const std::size_t codeword_size = (std::size_t)1 << codeword_domain_dim_;
        // The reference subtracts without looking (ldt_reducer.tcc:34-42): with RS_extra_dimensions = 1 the constraint degree bound reaches the codeword size,
        // its proximity parameter wraps around and the query count that follows is meaningless.  Refused here.
        if (max_constraint_degree_bound_ + 1 >= codeword_size || max_tested_degree_bound_ + 1 >= codeword_size)
            throw std::invalid_argument("the degree bounds leave no room in the codeword domain (RS_extra_dimensions too small)");
        absolute_proximity_parameter_ = std::min(codeword_size - max_constraint_degree_bound_, codeword_size - max_tested_degree_bound_) - 1;   // ldt_reducer.tcc:34-42
        num_output_LDT_instances_ = repetitions((double)interactive_soundness_error_bits_, (double)codeword_domain_dim_ - fbits);         // :53-56
        std::size_t total_localization = 0;
        for (std::size_t l : localization_parameters_) total_localization += l;
        if (max_tested_degree_bound_ % ((std::size_t)1 << total_localization))
            throw std::invalid_argument("FRI only supports testing degree bounds that are a multiple of 2^{sum of localization parameters}.");
        const double delta = (double)absolute_proximity_parameter_ / (double)codeword_size;   // fri_ldt.tcc:83-106 (heuristic)
        fri_query_repetitions_ = repetitions((double)query_soundness_error_bits_, std::log2(1 - delta));
        const double per_interaction = std::log2((double)(((std::size_t)1 << localization_parameters_[0]) - 1)) - fbits;
        fri_interactive_repetitions_ = repetitions((double)interactive_soundness_error_bits_, per_interaction);
    }
};

// ---- virtual oracles: evaluated_contents on device vectors ------------------------------------------------------------------------
template<typename FieldT>
class fz_virtual_oracle : public virtual_oracle<FieldT> {                                    // r1cs_rs_iop.tcc:141-222: fw * Z_I + f_1v
    std::size_t primary_input_size_;
    field_subset<FieldT> input_variable_domain_, codeword_domain_;
    device_vector<FieldT> f1v_coefficients_;
    bool primary_input_set_ = false;
public:
    fz_virtual_oracle(std::size_t primary_input_size, const field_subset<FieldT> &input_variable_domain, const field_subset<FieldT> &codeword_domain)
        : primary_input_size_(primary_input_size), input_variable_domain_(input_variable_domain), codeword_domain_(codeword_domain)
    {
        if (input_variable_domain.num_elements() > codeword_domain.num_elements()) throw std::invalid_argument("Codeword domain must be bigger than the input variable domain.");
    }
    void set_primary_input(const std::vector<FieldT> &primary_input)
    {
        if (primary_input.size() != primary_input_size_) throw std::invalid_argument("Primary input size does not match the previously declared size.");
        std::vector<FieldT> f1v_evals(1, field_host<FieldT>::one());                          // :199-206
        f1v_evals.insert(f1v_evals.end(), primary_input.begin(), primary_input.end());
        f1v_coefficients_ = dev::IFFT<FieldT>(device_vector<FieldT>(device_array<FieldT>::from_host(f1v_evals)), input_variable_domain_);   // :207-210
        primary_input_set_ = true;
    }
    const device_vector<FieldT> &f1v_coefficients() const { return f1v_coefficients_; }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &constituents) const override { return evaluated_contents_over(codeword_domain_, constituents); }
    bool restrictable() const override { return true; }
    std::size_t smallest_window() const override { return input_variable_domain_.num_elements(); }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &constituents) const override
    {
        if (constituents.size() != 1) throw std::invalid_argument("fz_virtual_oracle has one constituent oracle.");
        if (!primary_input_set_) throw std::logic_error("Evaluation requested before primary_input is set.");
        const field_subset<FieldT> &I = input_variable_domain_;
        const device_vector<FieldT> f1v = dev::FFT<FieldT>(f1v_coefficients_, I.num_elements(), D);                     // :211-212
        const field_subset<FieldT> L = dist::local_domain(D);                                 // pointwise: this rank's part is a domain of its own
        device_vector<FieldT> out(L.num_elements());
        if (dev::additive(L))
            check(iopx_fz_gf192_dev(constituents[0].words(), f1v.words(), dev::basis_words(L), L.dimension(), dev::shift_words(L), dev::basis_words(I),
                                    I.dimension(), dev::shift_words(I), out.words()));
        else
            check(iopx_fz_fp3_dev(constituents[0].words(), f1v.words(), L.dimension(), dev::gen_words(L), dev::shift_words(L), I.dimension(), dev::shift_words(I),
                                  out.words()));
        return out;
    }
};

template<typename FieldT>
class rowcheck_ABC_virtual_oracle : public virtual_oracle<FieldT> {                           // common/rowcheck.tcc:5-88
    field_subset<FieldT> codeword_domain_, constraint_domain_;
public:
    rowcheck_ABC_virtual_oracle(const field_subset<FieldT> &codeword_domain, const field_subset<FieldT> &constraint_domain)
        : codeword_domain_(codeword_domain), constraint_domain_(constraint_domain) {}
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    std::size_t smallest_window() const override { return constraint_domain_.num_elements(); }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != 3) throw std::invalid_argument("rowcheck_ABC has three constituent oracles.");
        const field_subset<FieldT> L = dist::local_domain(D);
        const field_subset<FieldT> &H = constraint_domain_;
        device_vector<FieldT> out(L.num_elements());
        if (dev::additive(L))
            check(iopx_rowcheck_gf192_dev(c[0].words(), c[1].words(), c[2].words(), dev::basis_words(L), L.dimension(), dev::shift_words(L), H.dimension(),
                                          dev::shift_words(H), out.words()));
        else
            check(iopx_rowcheck_fp3_dev(c[0].words(), c[1].words(), c[2].words(), L.dimension(), dev::gen_words(L), dev::shift_words(L), H.dimension(),
                                        dev::shift_words(H), out.words()));
        return out;
    }
};

template<typename FieldT>
class multi_lincheck_virtual_oracle : public virtual_oracle<FieldT> {                         // basic_lincheck_aux.tcc:5-144
    field_subset<FieldT> codeword_domain_, constraint_domain_, variable_domain_, summation_domain_;
    const std::vector<sparse_matrix<FieldT>> *matrices_T_;
    std::vector<FieldT> r_Mz_;
    device_vector<FieldT> p_alpha_evals_;
public:
    multi_lincheck_virtual_oracle(const field_subset<FieldT> &L, const field_subset<FieldT> &C, const field_subset<FieldT> &V, const field_subset<FieldT> &S,
                                  const std::vector<sparse_matrix<FieldT>> *transposed_matrices)
        : codeword_domain_(L), constraint_domain_(C), variable_domain_(V), summation_domain_(S), matrices_T_(transposed_matrices) {}
    // :29-99 — alpha powers, p_alpha_prime (the powers at the constraint positions of the summation domain), p_alpha_ABC
    // (sum_m r_m M_m^T applied to the powers); the interpolation of :94-98 happens together with the extension of :112-118
    void set_challenge(const FieldT &alpha, const std::vector<FieldT> &r_Mz)
    {
        if (r_Mz.size() != matrices_T_->size()) throw std::invalid_argument("Not enough random linear combination coefficients were provided");
        r_Mz_ = r_Mz;
        const field_subset<FieldT> &C = constraint_domain_, &S = summation_domain_;
        const device_vector<FieldT> alpha_powers = dev::pow_table<FieldT>(C.num_elements(), alpha, field_host<FieldT>::one());      // :37-45
        p_alpha_evals_ = device_vector<FieldT>(2 * S.num_elements());
        const device_vector<FieldT> prime = p_alpha_evals_.slice(0, S.num_elements()), abc = p_alpha_evals_.slice(S.num_elements(), S.num_elements());
        if (C.num_elements() == S.num_elements()) {
            prime.copy_from(alpha_powers);                                                   // reindex_by_subset is the identity
        } else {                                                                             // :50-58
            std::vector<uint64_t> idx(C.num_elements());
            for (std::size_t i = 0; i < idx.size(); ++i) idx[i] = S.reindex_by_subset(C.dimension(), i);
            const device_array<uint64_t> d_idx = device_array<uint64_t>::from_host(idx);
            prime.fill_zero();
            check(iopx_scatter_dev(alpha_powers.data(), d_idx.data(), idx.size(), sizeof(FieldT), prime.data()));
        }
        for (std::size_t m = 0; m < matrices_T_->size(); ++m) (*matrices_T_)[m].times_vector(alpha_powers, abc, &r_Mz_[m], m > 0);     // :64-88
    }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    std::size_t smallest_window() const override { return summation_domain_.num_elements(); }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != matrices_T_->size() + 1) throw std::invalid_argument("multi_lincheck uses more constituent oracles than what was provided.");
        const std::vector<device_vector<FieldT>> p = dev::reextend_packed<FieldT>(p_alpha_evals_, 2, summation_domain_, D);                    // :94-98 + :112-118
        const std::vector<const void *> Mz = dev::pointers(c, 1);
        device_vector<FieldT> out(c[0].size());
        auto fn = field_host<FieldT>::additive() ? iopx_lincheck_gf192_dev : iopx_lincheck_fp3_dev;
        check(fn(c[0].words(), Mz.data(), Mz.size(), detail::words(r_Mz_.data()), p[0].words(), p[1].words(), c[0].size(), out.words()));
        return out;
    }
};

template<typename FieldT>
class random_linear_combination_oracle : public virtual_oracle<FieldT> {                      // common/random_linear_combination.tcc
    std::size_t num_oracles_;
    std::vector<FieldT> coefficients_;
public:
    explicit random_linear_combination_oracle(std::size_t num_oracles) : num_oracles_(num_oracles) {}
    void set_random_coefficients(const std::vector<FieldT> &coefficients)
    {
        if (coefficients.size() != num_oracles_) throw std::invalid_argument("Random Linear Combination Oracle: Expected same number of random coefficients as oracles.");
        coefficients_ = coefficients;
    }
    bool restrictable() const override { return true; }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &, const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents(c); }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != num_oracles_) throw std::invalid_argument("Random Linear Combination Oracle: Expected same number of evaluations as in registration.");
        const std::vector<const void *> ptrs = dev::pointers(c);
        device_vector<FieldT> out(c[0].size());
        auto fn = field_host<FieldT>::additive() ? iopx_lincomb_gf192_dev : iopx_lincomb_fp3_dev;
        check(fn(ptrs.data(), ptrs.size(), detail::words(coefficients_.data()), c[0].size(), out.words()));
        return out;
    }
};

template<typename FieldT>
class sumcheck_g_oracle : public virtual_oracle<FieldT> {                                     // sumcheck.tcc:11-119
    field_subset<FieldT> summation_domain_, codeword_domain_;
    FieldT claimed_sum_;
public:
    sumcheck_g_oracle(const field_subset<FieldT> &summation_domain, const field_subset<FieldT> &codeword_domain)
        : summation_domain_(summation_domain), codeword_domain_(codeword_domain), claimed_sum_(field_host<FieldT>::zero()) {}
    void set_claimed_sum(const FieldT &claimed_sum) { claimed_sum_ = claimed_sum; }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    std::size_t smallest_window() const override { return summation_domain_.num_elements(); }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != 2) throw std::invalid_argument("sumcheck_g_oracle has two constituent oracles");
        const field_subset<FieldT> L = dist::local_domain(D);
        const field_subset<FieldT> &H = summation_domain_;
        device_vector<FieldT> out(L.num_elements());
        if (dev::additive(L))
            check(iopx_sumcheck_g_gf192_dev(c[0].words(), c[1].words(), dev::basis_words(L), L.dimension(), dev::shift_words(L), dev::basis_words(H), H.dimension(),
                                            dev::shift_words(H), detail::words(&claimed_sum_), out.words()));
        else
            check(iopx_sumcheck_g_fp3_dev(c[0].words(), c[1].words(), L.dimension(), dev::gen_words(L), dev::shift_words(L), H.dimension(), dev::shift_words(H),
                                          detail::words(&claimed_sum_), out.words()));
        return out;
    }
};

// combined_LDT_virtual_oracle on device vectors (ldt_reducer_aux.tcc:3-131); libiop_amd.hpp holds the host-vector form of the same class
template<typename FieldT>
class combined_LDT_device_oracle : public virtual_oracle<FieldT> {
    field_subset<FieldT> codeword_domain_;
    std::vector<std::size_t> degrees_;
    std::vector<FieldT> coefficients_;
public:
    combined_LDT_device_oracle(const field_subset<FieldT> &codeword_domain, const std::vector<std::size_t> &input_oracle_degrees)
        : codeword_domain_(codeword_domain), degrees_(input_oracle_degrees) {}
    void set_random_coefficients(const std::vector<FieldT> &coefficients)
    {
        if (coefficients.size() != 2 * degrees_.size()) throw std::invalid_argument("Expected the nunmber of random coefficients to be twice the number of oracles.");
        coefficients_ = coefficients;
    }
    device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &c) const override { return evaluated_contents_over(codeword_domain_, c); }
    bool restrictable() const override { return true; }
    device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &D, const std::vector<device_vector<FieldT>> &c) const override
    {
        if (c.size() != degrees_.size()) throw std::invalid_argument("Expected same number of evaluations as in registration.");
        const field_subset<FieldT> L = dist::local_domain(D);
        const std::vector<const void *> ptrs = dev::pointers(c);
        device_vector<FieldT> out(L.num_elements());
        if (dev::additive(L))
            check(iopx_ldt_combine_gf192_dev(ptrs.data(), ptrs.size(), degrees_.data(), detail::words(coefficients_.data()), dev::basis_words(L), L.dimension(),
                                             dev::shift_words(L), out.words()));
        else
            check(iopx_ldt_combine_fp3_dev(ptrs.data(), ptrs.size(), degrees_.data(), detail::words(coefficients_.data()), L.dimension(), dev::gen_words(L),
                                           dev::shift_words(L), out.words()));
        return out;
    }
};

// ---- protocols ------------------------------------------------------------------------------------------------------------------
template<typename FieldT>
class batch_sumcheck_protocol {                                                               // sumcheck.tcc:167-430, non-zk
    bcs_prover<FieldT> &IOP_;
    domain_handle summation_domain_handle_, codeword_domain_handle_;
    std::size_t degree_bound_, g_degree_, h_degree_;
    field_subset<FieldT> H_, L_;
    std::vector<oracle_handle> oracle_handles_;
    std::vector<FieldT> claimed_sums_;
    bool registered_ = false;
    verifier_random_message_handle challenge_handle_;
    oracle_handle h_handle_, combined_f_handle_, g_handle_;
    std::shared_ptr<random_linear_combination_oracle<FieldT>> combined_f_oracle_;
    std::shared_ptr<sumcheck_g_oracle<FieldT>> g_oracle_;
public:
    batch_sumcheck_protocol(bcs_prover<FieldT> &IOP, const domain_handle &summation_domain_handle, const domain_handle &codeword_domain_handle, std::size_t degree_bound)
        : IOP_(IOP), summation_domain_handle_(summation_domain_handle), codeword_domain_handle_(codeword_domain_handle), degree_bound_(degree_bound),
          H_(IOP.get_domain(summation_domain_handle)), L_(IOP.get_domain(codeword_domain_handle))
    {
        g_degree_ = H_.num_elements() - 1;
        h_degree_ = degree_bound - H_.num_elements();
    }
    void attach_oracle_for_summing(const oracle_handle &handle, const FieldT &claimed_sum = field_host<FieldT>::zero())
    {
        if (registered_) throw std::logic_error("Called attach_oracle_for_summing after register_proof.");
        oracle_handles_.push_back(handle);
        claimed_sums_.push_back(claimed_sum);
    }
    void register_challenge() { challenge_handle_ = IOP_.register_verifier_random_message(oracle_handles_.size()); }       // :199-206
    void register_proof()                                                                    // :235-273
    {
        h_handle_ = IOP_.register_oracle("sumcheck h", codeword_domain_handle_, h_degree_, false);
        combined_f_oracle_ = std::make_shared<random_linear_combination_oracle<FieldT>>(oracle_handles_.size());
        combined_f_handle_ = IOP_.register_virtual_oracle(codeword_domain_handle_, degree_bound_, oracle_handles_, combined_f_oracle_, true);
        g_oracle_ = std::make_shared<sumcheck_g_oracle<FieldT>>(H_, L_);
        g_handle_ = IOP_.register_virtual_oracle(codeword_domain_handle_, g_degree_, { combined_f_handle_, h_handle_ }, g_oracle_);
        registered_ = true;
    }
    void calculate_and_submit_proof()                                                        // :343-388
    {
        const std::vector<FieldT> challenge = IOP_.obtain_verifier_random_message(challenge_handle_);
        combined_f_oracle_->set_random_coefficients(challenge);
        FieldT combined_claimed_sum = field_host<FieldT>::zero();                            // :327-341
        for (std::size_t i = 0; i < challenge.size(); ++i)
            combined_claimed_sum = field_host<FieldT>::add(combined_claimed_sum, field_host<FieldT>::mul(challenge[i], claimed_sums_[i]));
        g_oracle_->set_claimed_sum(combined_claimed_sum);
        device_vector<FieldT> h;
        const std::size_t count = (std::size_t)1 << detail::log2_ceil(degree_bound_);
        if (degree_bound_ > H_.num_elements() && dev::use_head(L_, count) && IOP_.can_restrict(combined_f_handle_)) {
            // the interpolant of :351-354 only reads the head of the codeword domain: evaluate the combined f there and nowhere else
            h = device_vector<FieldT>(degree_bound_ - H_.num_elements());
            // Subspaces, f over exactly two cosets C0, C1 of H (the usual case: deg f < 2 |H|): f = Z_H h + g with deg g, deg h < |H| and Z_H constant
            // on a coset of H (z0, z1), so with R = the re-extension from one coset to the other (exact on polynomials of fewer than |H| coefficients)
            // (C1 to C0 here) f|C0 - R(f|C1) = (z0 - z1) h|C0: h over C0 comes from one re-extension and one scaling, and its codeword from re-extending
            // THAT (its first |H| entries are h over C0 itself) — no coefficient form, no division pass.  The quotient is unique, so these are :359-365's field elements.
            const bool two_cosets = dev::additive(L_) && count == 2 * H_.num_elements() && dev::spanned_by_prefix(H_, L_);
            if (two_cosets) h = device_vector<FieldT>(H_.num_elements());
            if (!L_.distributed() || dist::ctx().rank == 0) {
                const dist::one_rank_section alone;
                const std::size_t later = IOP_.head_hint();         // the LDT reads the (cached) combined f over its own head: evaluate that one if it is larger
                if (later > count && dev::use_head(L_, later)) (void)IOP_.get_oracle_evaluations_over_head(combined_f_handle_, later);
                const device_vector<FieldT> evals = IOP_.get_oracle_evaluations_over_head(combined_f_handle_, count);
                if (two_cosets) {
                    const std::size_t n = H_.num_elements();
                    const field_subset<FieldT> C0 = dist::window_domain(L_, dist::window{ 0, 1, n }), C1 = dist::window_domain(L_, dist::window{ n, 1, n });
                    const device_vector<FieldT> moved = dev::reextend_packed<FieldT>(evals.slice(n, n), 1, C1, C0)[0];
                    typedef field_host<FieldT> F;
                    const FieldT scale = F::inverse(F::sub(F::vanishing_eval(H_, C0.shift()), F::vanishing_eval(H_, C1.shift())));
                    const std::vector<device_vector<FieldT>> terms = { evals.slice(0, n), moved };
                    const std::vector<const void *> ptrs = dev::pointers(terms);
                    const FieldT coefficients[2] = { scale, F::neg(scale) };
                    check(iopx_lincomb_gf192_dev(ptrs.data(), 2, detail::words(coefficients), n, h.words()));
                } else {
                    h = dev::poly_div_vanishing<FieldT>(dev::IFFT<FieldT>(evals, dist::head_domain(L_, count)), degree_bound_, H_);
                }
            }
            if (L_.distributed()) dist::broadcast<FieldT>(h, 0);
            if (two_cosets) {                                                                                               // h over C0 -> h over L
                IOP_.submit_oracle(h_handle_, oracle<FieldT>(dev::reextend_packed<FieldT>(h, 1, dist::window_domain(L_, dist::window{ 0, 1, H_.num_elements() }), L_, true)[0]));
                return;
            }
        } else {
            const device_vector<FieldT> evals = IOP_.get_oracle_evaluations(combined_f_handle_);
            h = dev::interpolate_and_divide<FieldT>(evals, degree_bound_, L_, H_);                                         // :351-354, :359-365
        }
        IOP_.submit_oracle(h_handle_, oracle<FieldT>(dev::FFT<FieldT>(h, h.size(), L_)));                                  // :384-387
    }
    std::vector<oracle_handle> get_all_oracle_handles() const { return { h_handle_, g_handle_ }; }
};

template<typename FieldT>
class multi_lincheck {                                                                        // basic_lincheck.tcc:113-296, non-zk
    bcs_prover<FieldT> &IOP_;
    domain_handle codeword_domain_handle_, summation_domain_handle_;
    std::size_t repetitions_, num_matrices_, lincheck_degree_;
    std::vector<oracle_handle> constituent_oracle_handles_;
    std::vector<std::shared_ptr<batch_sumcheck_protocol<FieldT>>> sumchecks_;
    std::vector<std::shared_ptr<multi_lincheck_virtual_oracle<FieldT>>> oracles_;
    std::vector<verifier_random_message_handle> alpha_handles_, random_coefficient_handles_;
public:
    multi_lincheck(bcs_prover<FieldT> &IOP, const domain_handle &codeword_domain_handle, const domain_handle &constraint_domain_handle,
                   const domain_handle &variable_domain_handle, const std::vector<sparse_matrix<FieldT>> *transposed_matrices, const oracle_handle &fz_handle,
                   const std::vector<oracle_handle> &Mz_handles, std::size_t repetitions)
        : IOP_(IOP), codeword_domain_handle_(codeword_domain_handle), repetitions_(repetitions), num_matrices_(transposed_matrices->size())
    {
        if (num_matrices_ < 1) throw std::invalid_argument("multi_lincheck expects at least one matrix");
        if (Mz_handles.size() != num_matrices_) throw std::invalid_argument("inconsistent number of Mz_handles and matrices passed into multi lincheck.");
        const field_subset<FieldT> L = IOP.get_domain(codeword_domain_handle), C = IOP.get_domain(constraint_domain_handle), V = IOP.get_domain(variable_domain_handle);
        summation_domain_handle_ = C.dimension() > V.dimension() ? constraint_domain_handle : variable_domain_handle;      // :137-143
        const field_subset<FieldT> S = IOP.get_domain(summation_domain_handle_);
        constituent_oracle_handles_.push_back(fz_handle);
        constituent_oracle_handles_.insert(constituent_oracle_handles_.end(), Mz_handles.begin(), Mz_handles.end());
        lincheck_degree_ = S.num_elements() + std::max(IOP.get_oracle_degree(fz_handle), IOP.get_oracle_degree(Mz_handles[0])) - 1;   // :151-154
        for (std::size_t i = 0; i < repetitions; ++i) {
            sumchecks_.push_back(std::make_shared<batch_sumcheck_protocol<FieldT>>(IOP, summation_domain_handle_, codeword_domain_handle, lincheck_degree_));
            oracles_.push_back(std::make_shared<multi_lincheck_virtual_oracle<FieldT>>(L, C, V, S, transposed_matrices));
        }
    }
    void register_challenge()                                                                // :197-218
    {
        for (std::size_t i = 0; i < repetitions_; ++i) alpha_handles_.push_back(IOP_.register_verifier_random_message(1));
        for (std::size_t i = 0; i < repetitions_; ++i) random_coefficient_handles_.push_back(IOP_.register_verifier_random_message(num_matrices_));
        for (std::size_t i = 0; i < repetitions_; ++i) {
            const oracle_handle h = IOP_.register_virtual_oracle(codeword_domain_handle_, lincheck_degree_, constituent_oracle_handles_, oracles_[i]);
            sumchecks_[i]->attach_oracle_for_summing(h);
            sumchecks_[i]->register_challenge();
        }
    }
    void register_proof() { for (auto &s : sumchecks_) s->register_proof(); }
    void calculate_and_submit_proof()                                                        // :241-257
    {
        for (std::size_t i = 0; i < repetitions_; ++i) {
            const FieldT alpha = IOP_.obtain_verifier_random_message(alpha_handles_[i])[0];
            const std::vector<FieldT> r_Mz = IOP_.obtain_verifier_random_message(random_coefficient_handles_[i]);
            oracles_[i]->set_challenge(alpha, r_Mz);
            sumchecks_[i]->calculate_and_submit_proof();
        }
    }
    std::vector<oracle_handle> get_all_oracle_handles() const
    {
        std::vector<oracle_handle> out;
        for (auto &s : sumchecks_) for (auto &h : s->get_all_oracle_handles()) out.push_back(h);
        return out;
    }
};

template<typename FieldT>
class encoded_aurora_protocol {                                                               // r1cs_rs_iop.tcc:252-693, non-zk
    bcs_prover<FieldT> &IOP_;
    const r1cs_constraint_system<FieldT> &cs_;
    field_subset<FieldT> C_, V_, L_, I_;
    oracle_handle fw_handle_, fAz_handle_, fBz_handle_, fCz_handle_, fz_handle_, rowcheck_handle_;
    std::shared_ptr<fz_virtual_oracle<FieldT>> fz_oracle_;
    std::shared_ptr<rowcheck_ABC_virtual_oracle<FieldT>> rowcheck_oracle_;
    std::shared_ptr<multi_lincheck<FieldT>> multi_lincheck_;
    const std::vector<sparse_matrix<FieldT>> *transposed_matrices_ = nullptr;
public:
    encoded_aurora_protocol(bcs_prover<FieldT> &IOP, const domain_handle &constraint_domain_handle, const domain_handle &variable_domain_handle,
                            const domain_handle &codeword_domain_handle, const r1cs_constraint_system<FieldT> &constraint_system, std::size_t lincheck_repetitions,
                            bool holographic = false)                                        // holographic: the lincheck is fractal.hpp's (r1cs_rs_iop.tcc:344-357)
        : IOP_(IOP), cs_(constraint_system), C_(IOP.get_domain(constraint_domain_handle)), V_(IOP.get_domain(variable_domain_handle)),
          L_(IOP.get_domain(codeword_domain_handle))
    {
        typedef aurora_snark_parameters<FieldT> P;
        if (!P::is_pow2(cs_.num_inputs() + 1))
            throw std::invalid_argument("number of inputs in the constraint system must be one less than a power of two.Perhaps pad your number of inputs");
        I_ = V_.get_subset_of_order(cs_.num_inputs() + 1);                                   // :279-280
        // register_witness_oracles (:285-375), query bound 0
        const std::size_t m = (std::size_t)1 << detail::log2_ceil(cs_.num_constraints()), n = (std::size_t)1 << detail::log2_ceil(cs_.num_variables()), k = cs_.num_inputs();
        const std::size_t fw_degree = n - (k + 1);
        fw_handle_ = IOP.register_oracle("fw", codeword_domain_handle, fw_degree, false);
        fAz_handle_ = IOP.register_oracle("fAz", codeword_domain_handle, m, false);
        fBz_handle_ = IOP.register_oracle("fBz", codeword_domain_handle, m, false);
        fCz_handle_ = IOP.register_oracle("fCz", codeword_domain_handle, m, false);
        fz_oracle_ = std::make_shared<fz_virtual_oracle<FieldT>>(k, I_, L_);
        fz_handle_ = IOP.register_virtual_oracle(codeword_domain_handle, fw_degree + k + 1, { fw_handle_ }, fz_oracle_);
        const std::vector<oracle_handle> Mz_handles = { fAz_handle_, fBz_handle_, fCz_handle_ };
        // the matrices as set_challenge walks them: column c of M lands at summation index reindex(reindex(c)) (basic_lincheck_aux.tcc:80-84)
        const field_subset<FieldT> &S = C_.dimension() > V_.dimension() ? C_ : V_;
        const std::string key = std::to_string((int)S.type()) + ":" + std::to_string(S.dimension()) + ":" + std::to_string(V_.dimension()) + ":" + std::to_string(I_.dimension());
        auto it = cs_.lincheck_matrix_cache_.find(key);
        if (it == cs_.lincheck_matrix_cache_.end()) {
            const auto t_cold = std::chrono::steady_clock::now();
            struct cold_done { std::chrono::steady_clock::time_point t; ~cold_done() { (void)iopx_cold_add("transposed lincheck matrices", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); } } cold_{ t_cold };
            std::vector<std::size_t> col_to_summation(cs_.num_variables() + 1);
            for (std::size_t c = 0; c < col_to_summation.size(); ++c) col_to_summation[c] = S.reindex_by_subset(V_.dimension(), V_.reindex_by_subset(I_.dimension(), c));
            std::vector<sparse_matrix<FieldT>> T;
            T.push_back(cs_.A.transposed_onto(S.num_elements(), col_to_summation));
            T.push_back(cs_.B.transposed_onto(S.num_elements(), col_to_summation));
            T.push_back(cs_.C.transposed_onto(S.num_elements(), col_to_summation));
            it = cs_.lincheck_matrix_cache_.emplace(key, std::move(T)).first;
        }
        transposed_matrices_ = &it->second;
        if (!holographic)
            multi_lincheck_ = std::make_shared<multi_lincheck<FieldT>>(IOP, codeword_domain_handle, constraint_domain_handle, variable_domain_handle, &it->second,
                                                                       fz_handle_, Mz_handles, lincheck_repetitions);
        rowcheck_oracle_ = std::make_shared<rowcheck_ABC_virtual_oracle<FieldT>>(L_, C_);
        rowcheck_handle_ = IOP.register_virtual_oracle(codeword_domain_handle, C_.num_elements() - 1, Mz_handles, rowcheck_oracle_);
    }
    void register_challenge() { multi_lincheck_->register_challenge(); }
    void register_proof() { multi_lincheck_->register_proof(); }
    const field_subset<FieldT> &input_variable_domain() const { return I_; }
    const std::vector<sparse_matrix<FieldT>> *transposed_matrices() const { return transposed_matrices_; }
    oracle_handle fz_handle() const { return fz_handle_; }
    std::vector<oracle_handle> Mz_handles() const { return { fAz_handle_, fBz_handle_, fCz_handle_ }; }
    std::vector<oracle_handle> witness_and_rowcheck_handles() const { return { fw_handle_, fAz_handle_, fBz_handle_, fCz_handle_, rowcheck_handle_ }; }

    // :481-615.  f_w' interpolates z - f_1v over the variable domain (zero on the input positions, where f_1v already equals z), is
    // divided by Z_I and extended together with f_Az, f_Bz, f_Cz.  d_assignment: the variable assignment (1, primary, auxiliary)
    // already resident in HBM (then auxiliary_input is not read).
    void submit_witness_oracles(const std::vector<FieldT> &primary_input, const std::vector<FieldT> &auxiliary_input, const device_vector<FieldT> *d_assignment = nullptr)
    {
        cs_.prepare_device();
        fz_oracle_->set_primary_input(primary_input);                                        // :485, :508-516
        const device_vector<FieldT> f1v_over_variable_domain = dev::FFT<FieldT>(fz_oracle_->f1v_coefficients(), I_.num_elements(), V_);      // :517-518
        device_vector<FieldT> d_z;
        if (d_assignment) d_z = *d_assignment;
        else {                                                                               // :581-585
            std::vector<FieldT> z(1, field_host<FieldT>::one());
            z.insert(z.end(), primary_input.begin(), primary_input.end());
            z.insert(z.end(), auxiliary_input.begin(), auxiliary_input.end());
            d_z = device_vector<FieldT>(device_array<FieldT>::from_host(z));
        }
        if (d_z.size() != cs_.num_variables() + 1) throw std::invalid_argument("variable assignment of the wrong size");
        device_vector<FieldT> z_over_variable_domain = d_z;
        if (!dev::additive(V_)) {                                                            // create_fw_prime_evals' reindexing (:421-423)
            const std::string key = "variable order:" + std::to_string(V_.dimension()) + ":" + std::to_string(I_.dimension());
            auto it = cs_.index_cache_.find(key);
            if (it == cs_.index_cache_.end()) {
                std::vector<uint64_t> order(V_.num_elements());
                for (std::size_t i = 0; i < order.size(); ++i) order[V_.reindex_by_subset(I_.dimension(), i)] = i;
                it = cs_.index_cache_.emplace(key, device_array<uint64_t>::from_host(order)).first;
            }
            z_over_variable_domain = device_vector<FieldT>(V_.num_elements());
            check(iopx_gather_dev(d_z.data(), it->second.data(), V_.num_elements(), sizeof(FieldT), z_over_variable_domain.data()));
        }
        const device_vector<FieldT> fw_prime_evals = dev::sub<FieldT>(z_over_variable_domain, f1v_over_variable_domain);                     // :406-430
        const std::size_t nC = C_.num_elements();
        const device_vector<FieldT> Mz(3 * nC);
        if (cs_.num_constraints() != nC) Mz.fill_zero();
        const sparse_matrix<FieldT> *M[3] = { &cs_.A, &cs_.B, &cs_.C };
        for (int q = 0; q < 3; ++q) M[q]->times_vector(d_z, Mz.slice(q * nC, cs_.num_constraints()));                                       // :586-592, r1cs.tcc:236-268
        std::vector<device_vector<FieldT>> codewords;
        if (dev::head_evaluation_enabled() && dev::spanned_by_prefix(V_, L_) && dev::spanned_by_prefix(I_, V_) && V_.dimension() < L_.dimension() && I_.num_elements() < V_.num_elements()) {
            // f_w = f_w' / Z_I exactly (f_w' vanishes on the input positions by construction), deg f_w < |V|: so f_w over the first coset V0 of V inside L
            // is f_w' there (one re-extension from V) divided by Z_I pointwise (V0 does not meet I), and its codeword is the re-extension of THAT —
            // the coefficient form (:551-555), the division passes (:563-565) and both basis conversions drop out.  Same field elements.
            field_subset<FieldT> whole = L_;
            whole.set_distributed(false);
            const field_subset<FieldT> V0 = dist::window_domain(whole, dist::window{ 0, 1, V_.num_elements() });
            const device_vector<FieldT> fw_prime_V0 = dev::reextend_packed<FieldT>(fw_prime_evals, 1, V_, V0)[0];
            const device_vector<FieldT> fw_V0(V0.num_elements());                            // Z_I is constant on the cosets of I: one product per element
            check(iopx_div_by_vanishing_gf192_dev(fw_prime_V0.words(), dev::basis_words(V0), V0.dimension(), dev::shift_words(V0), I_.dimension(), dev::shift_words(I_),
                                                  fw_V0.words()));
            if (V0.dimension() == C_.dimension()) {
                // one batch of four: at the pair bits where three polynomials leave a wavefront a quarter empty, four fill it
                const std::vector<device_vector<FieldT>> four = dev::reextend2_packed<FieldT>(Mz, 3, C_, fw_V0, 1, V0, L_);
                codewords = { four[3], four[0], four[1], four[2] };
            } else {
                codewords.push_back(dev::reextend_packed<FieldT>(fw_V0, 1, V0, L_, true)[0]);
                for (auto &cw : dev::reextend_packed<FieldT>(Mz, 3, C_, L_)) codewords.push_back(cw);
            }
        } else {
            const device_vector<FieldT> fw_prime = dev::IFFT<FieldT>(fw_prime_evals, V_);                                                    // :551-555
            const device_vector<FieldT> fw = dev::poly_div_vanishing<FieldT>(fw_prime, V_.num_elements(), I_);                               // :563-565
            codewords = dev::FFT_and_reextend_packed<FieldT>(fw, Mz, 3, C_, L_);                                                             // :567-568, :459-478
        }
        const oracle_handle handles[4] = { fw_handle_, fAz_handle_, fBz_handle_, fCz_handle_ };
        for (int q = 0; q < 4; ++q) IOP_.submit_oracle(handles[q], oracle<FieldT>(codewords[q]));                                           // :603-606
    }
    void calculate_and_submit_proof() { multi_lincheck_->calculate_and_submit_proof(); }
    std::vector<oracle_handle> get_all_oracle_handles() const                                // :651-672
    {
        std::vector<oracle_handle> out = multi_lincheck_->get_all_oracle_handles();
        for (const oracle_handle &h : { fw_handle_, fAz_handle_, fBz_handle_, fCz_handle_, rowcheck_handle_ }) out.push_back(h);
        return out;
    }
};

template<typename FieldT>
class FRI_protocol {                                                                          // fri_ldt.tcc:260-548
    bcs_prover<FieldT> &IOP_;
    domain_handle codeword_domain_handle_;
    std::vector<oracle_handle> poly_handles_;
    std::vector<std::size_t> localization_;
    std::size_t poly_degree_bound_, interactive_repetitions_, query_repetitions_, num_reductions_, final_polynomial_degree_bound_ = 0;
    std::vector<field_subset<FieldT>> domains_;
    std::vector<domain_handle> domain_handles_;
    std::vector<std::vector<std::vector<oracle_handle>>> oracle_handles_;                    // [round][interaction][ldt]
    std::vector<std::vector<verifier_random_message_handle>> verifier_challenge_handles_;
    std::vector<std::vector<prover_message_handle>> final_polynomial_handles_;

    void compute_domains()                                                                   // :279-340
    {
        const field_subset<FieldT> L = IOP_.get_domain(codeword_domain_handle_);
        domains_.push_back(L);
        if (dev::additive(L)) {                                                              // :310-338 through the library's host-side helper
            std::size_t total = 0, d = L.dimension();
            std::vector<std::size_t> dims;
            for (std::size_t eta : localization_) { d -= eta; dims.push_back(d); total += d; }
            std::vector<FieldT> bases(total ? total : 1), shifts(localization_.size());
            check(iopx_fri_domains_gf192(dev::basis_words(L), L.dimension(), dev::shift_words(L), localization_.data(), localization_.size(),
                                         detail::words(bases.data()), detail::words(shifts.data())));
            std::size_t off = 0;
            for (std::size_t i = 0; i < dims.size(); ++i) {
                domains_.push_back(field_subset<FieldT>(affine_subspace<FieldT>(std::vector<FieldT>(bases.begin() + off, bases.begin() + off + dims[i]), shifts[i])));
                off += dims[i];
            }
        } else {                                                                             // :292-308: size >>= eta, shift <- shift^(2^eta)
            FieldT sh = L.shift();
            std::size_t logn = L.dimension();
            for (std::size_t eta : localization_) {
                sh = field_host<FieldT>::pow(sh, (uint64_t)1 << eta);
                logn -= eta;
                domains_.push_back(field_subset<FieldT>((std::size_t)1 << logn, sh));
            }
        }
        dist::mark_fri_domains(domains_, localization_);                                    // which L^(i) stay split over the ranks (dist.hpp)
    }
    // f_1 without f_0 over all of L.  A virtual f_0 is only ever folded; its polynomial has at most `head` coefficients (2^ceil(log2 of the tested
    // degree bound)), so f_1 = fold(f_0, x_0) has head / 2^eta_0 and is determined by its values on the head of L^(1) — which are the fold of f_0's
    // values on the head of L (FRI's cosets of L, subspace.tcc:73-91 / subgroup.tcc:175-197, are cosets of the head too).  So: the virtual oracles
    // over the head only (|L| / head times fewer evaluations), one fold there, interpolate, extend over L^(1): the same field elements as folding
    // all of f_0 (:522-526) WHEN f_0 is that polynomial.  For an instance whose virtual oracles are not polynomials (an unsatisfied witness, a
    // mismatched index: the reference still emits a transcript, which its verifier rejects) the two routes differ, so the result is confirmed on a
    // second window next to the head (as small as the virtual oracles can be evaluated over) — fold(f_0) there must be the extension's values — and on any difference this returns false
    // and the caller takes the reference's route.  A rational function that is not the polynomial differs from it on all but boundedly many points,
    // so a whole window of agreement does not happen by accident.
    bool first_round_from_head(std::vector<std::vector<device_vector<FieldT>>> &by_interaction)
    {
        if (num_reductions_ < 2 || !dev::head_evaluation_enabled()) return false;
        const field_subset<FieldT> &L0 = domains_[0], &L1 = domains_[1];
        const std::size_t head = (std::size_t)1 << detail::log2_ceil(poly_degree_bound_), cs0 = (std::size_t)1 << localization_[0];
        if (!dev::use_head(L0, head) || head / cs0 < 2) return false;
        for (auto &h : poly_handles_) if (!h.is_virtual || !IOP_.can_restrict(h)) return false;
        const dist::window h0 = dist::head_window(L0, head), h1 = dist::head_window(L1, head / cs0);
        std::size_t confirm = 2 * cs0;                                                        // the confirmation window: as small as the virtual oracles allow
        for (auto &h : poly_handles_) confirm = std::max(confirm, IOP_.smallest_window(h));
        confirm = std::min(head, (std::size_t)1 << detail::log2_ceil(confirm));
        const dist::window c0 = dist::beside_head_window(L0, head, confirm), c1 = dist::beside_head_window(L1, head / cs0, confirm / cs0);  // c0 folds onto c1
        const std::size_t me = dist::ctx().rank, checker = dist::window_owner(L0, c0);
        if (checker == (std::size_t)-1 || (L1.distributed() && dist::window_owner(L1, c1) != checker)) return false;
        const bool split = L0.distributed();
        const field_subset<FieldT> D_h0 = dist::window_domain(L0, h0), D_h1 = dist::window_domain(L1, h1), D_c0 = dist::window_domain(L0, c0);
        const device_array<uint64_t> d_differ(1);
        check(iopx_memset_dev(d_differ.data(), 0, 8));
        by_interaction.assign(interactive_repetitions_, std::vector<device_vector<FieldT>>(poly_handles_.size()));
        for (std::size_t l = 0; l < poly_handles_.size(); ++l) {
            device_vector<FieldT> f0;
            if (!split || me == 0) { const dist::one_rank_section alone; f0 = IOP_.get_oracle_evaluations_over_window(poly_handles_[l], h0); }
            for (std::size_t j = 0; j < interactive_repetitions_; ++j) {
                const FieldT x_0 = IOP_.obtain_verifier_random_message(verifier_challenge_handles_[0][j])[0];
                device_vector<FieldT> f1_head(head / cs0);                                         // f_1 over the head of L^(1)
                if (!split || me == 0) { const dist::one_rank_section alone; f1_head = dev::fold<FieldT>(f0, D_h0, cs0, x_0); }
                if (split) dist::broadcast<FieldT>(f1_head, 0);
                by_interaction[j][l] = dev::reextend_packed<FieldT>(f1_head, 1, D_h1, L1)[0];     // subspaces: no coefficient form in between
            }
            if (!split || me == checker) {
                const dist::one_rank_section alone;
                const device_vector<FieldT> f0c = IOP_.get_oracle_evaluations_over_window(poly_handles_[l], c0);
                for (std::size_t j = 0; j < interactive_repetitions_; ++j) {
                    const FieldT x_0 = IOP_.obtain_verifier_random_message(verifier_challenge_handles_[0][j])[0];
                    const device_vector<FieldT> folded = dev::fold<FieldT>(f0c, D_c0, cs0, x_0), extended = dist::window_of<FieldT>(by_interaction[j][l], L1, c1);
                    check(iopx_count_mismatch_dev(folded.data(), extended.data(), folded.size() * sizeof(FieldT), d_differ.data()));
                }
            }
        }
        if (split) check(iopx_comm_all_reduce_u64_dev(dist::ctx().comm, d_differ.data(), 1, IOPX_COMM_SUM));
        uint64_t differ = 0;
        check(iopx_memcpy_d2h(&differ, d_differ.data(), 8));
        if (differ) by_interaction.clear();
        return differ == 0;
    }
public:
    FRI_protocol(bcs_prover<FieldT> &IOP, const domain_handle &codeword_domain_handle, const std::vector<oracle_handle> &poly_handles,
                 const std::vector<std::size_t> &localization_parameters, std::size_t poly_degree_bound, std::size_t interactive_repetitions, std::size_t query_repetitions)
        : IOP_(IOP), codeword_domain_handle_(codeword_domain_handle), poly_handles_(poly_handles), localization_(localization_parameters),
          poly_degree_bound_(poly_degree_bound), interactive_repetitions_(interactive_repetitions), query_repetitions_(query_repetitions),
          num_reductions_(localization_parameters.size())
    {
        compute_domains();
    }
    // the head of L^(0) and the confirmation window beside it: the two windows first_round_from_head evaluates every oracle of the LDT over
    bool head_windows(dist::window &h0, dist::window &c0, std::size_t &confirm_out)
    {
        if (num_reductions_ < 2) return false;
        const field_subset<FieldT> &L0 = domains_[0];
        const std::size_t head = (std::size_t)1 << detail::log2_ceil(poly_degree_bound_), cs0 = (std::size_t)1 << localization_[0];
        if (head * 4 > L0.num_elements() || L0.num_elements() % head || head / cs0 < 2) return false;
        std::size_t confirm = 2 * cs0;
        for (auto &h : poly_handles_) confirm = std::max(confirm, IOP_.smallest_window(h));
        confirm = std::min(head, (std::size_t)1 << detail::log2_ceil(confirm));
        h0 = dist::head_window(L0, head);
        c0 = dist::beside_head_window(L0, head, confirm);
        confirm_out = confirm;
        return true;
    }
    void register_interactions()                                                             // :342-398
    {
        {
            dist::window h0, c0;
            std::size_t confirm;
            if (dev::head_evaluation_enabled() && head_windows(h0, c0, confirm)) { IOP_.want_window(codeword_domain_handle_, h0); IOP_.want_window(codeword_domain_handle_, c0); }
        }
        std::size_t total = localization_[0];
        domain_handles_.assign(num_reductions_, codeword_domain_handle_);
        oracle_handles_.assign(num_reductions_, {});
        oracle_handles_[0].push_back(poly_handles_);
        verifier_challenge_handles_.emplace_back();
        for (std::size_t j = 0; j < interactive_repetitions_; ++j) verifier_challenge_handles_[0].push_back(IOP_.register_verifier_random_message(1));
        for (std::size_t i = 1; i < num_reductions_; ++i) {
            total += localization_[i];
            const domain_handle L_i = IOP_.register_domain(domains_[i]);
            for (std::size_t j = 0; j < interactive_repetitions_; ++j) {
                oracle_handles_[i].emplace_back();
                for (std::size_t l = 0; l < poly_handles_.size(); ++l)
                    oracle_handles_[i][j].push_back(IOP_.register_oracle("f_" + std::to_string(i), L_i, poly_degree_bound_ >> total, false));
            }
            IOP_.set_round_parameters(domains_[i].get_subset_of_order((std::size_t)1 << localization_[i]));
            verifier_challenge_handles_.emplace_back();
            for (std::size_t j = 0; j < interactive_repetitions_; ++j) verifier_challenge_handles_[i].push_back(IOP_.register_verifier_random_message(1));
            domain_handles_[i] = L_i;
        }
        final_polynomial_degree_bound_ = poly_degree_bound_ >> total;
        for (std::size_t j = 0; j < interactive_repetitions_; ++j) {
            final_polynomial_handles_.emplace_back();
            for (std::size_t l = 0; l < poly_handles_.size(); ++l) final_polynomial_handles_[j].push_back(IOP_.register_prover_message(final_polynomial_degree_bound_));
        }
    }
    void register_queries()                                                                  // :400-472
    {
        for (std::size_t q = 0; q < query_repetitions_; ++q) {
            const query_position_handle s0 = IOP_.register_random_query_position(domain_handles_[0]);
            std::vector<std::vector<query_position_handle>> coset_positions(num_reductions_);
            {
                const field_subset<FieldT> d0 = domains_[0];
                const std::size_t cs0 = (std::size_t)1 << localization_[0];
                for (std::size_t i = 0; i < cs0; ++i)                                          // iop/utilities/query_positions.tcc
                    coset_positions[0].push_back(IOP_.register_deterministic_query_position(
                        { s0 }, [d0, cs0, i](const std::vector<std::size_t> &seed) { return d0.position_by_coset_indices(d0.coset_index(seed[0], cs0), i, cs0); }));
            }
            for (std::size_t r = 1; r < num_reductions_; ++r) {                               // fri_aux.tcc:351-387
                const field_subset<FieldT> prev = domains_[r - 1], cur = domains_[r];
                const std::size_t prev_cs = (std::size_t)1 << localization_[r - 1], cur_cs = (std::size_t)1 << localization_[r];
                for (std::size_t i = 0; i < cur_cs; ++i)
                    coset_positions[r].push_back(IOP_.register_deterministic_query_position(
                        { coset_positions[r - 1][0] }, [prev, cur, prev_cs, cur_cs, i](const std::vector<std::size_t> &seed) {
                            return cur.position_by_coset_indices(cur.coset_index(prev.coset_index(seed[0], prev_cs), cur_cs), i, cur_cs);
                        }));
            }
            for (std::size_t interaction = 0; interaction < interactive_repetitions_; ++interaction)
                for (std::size_t ldt = 0; ldt < poly_handles_.size(); ++ldt)
                    for (std::size_t r = 0; r < num_reductions_; ++r) {
                        const std::size_t queried_interaction = r == 0 ? 0 : interaction;
                        for (std::size_t j = 0; j < ((std::size_t)1 << localization_[r]); ++j)
                            IOP_.register_query(oracle_handles_[r][queried_interaction][ldt], coset_positions[r][j]);
                    }
        }
    }
    void calculate_and_submit_proof()                                                        // :474-548
    {
        std::vector<std::vector<device_vector<FieldT>>> by_interaction;
        const bool from_head = first_round_from_head(by_interaction);
        const std::size_t first_round = from_head ? 1 : 0;
        if (!from_head) {
            std::vector<device_vector<FieldT>> first;
            for (auto &h : poly_handles_) first.push_back(IOP_.get_oracle_evaluations(h));
            by_interaction.assign(interactive_repetitions_, first);
        }
        for (std::size_t i = first_round; i < num_reductions_; ++i) {
            const std::size_t cs = (std::size_t)1 << localization_[i];
            if (i > 0) {
                for (std::size_t j = 0; j < interactive_repetitions_; ++j)
                    for (std::size_t l = 0; l < poly_handles_.size(); ++l) IOP_.submit_oracle(oracle_handles_[i][j][l], oracle<FieldT>(by_interaction[j][l]));     // device-resident: no copy
                IOP_.signal_prover_round_done();
            }
            for (std::size_t j = 0; j < interactive_repetitions_; ++j) {
                const FieldT x_i = IOP_.obtain_verifier_random_message(verifier_challenge_handles_[i][j])[0];
                for (std::size_t l = 0; l < poly_handles_.size(); ++l) by_interaction[j][l] = dev::fold<FieldT>(by_interaction[j][l], domains_[i], cs, x_i, domains_[i + 1].distributed());   // :522-526
            }
        }
        for (std::size_t j = 0; j < interactive_repetitions_; ++j)
            for (std::size_t l = 0; l < poly_handles_.size(); ++l) {
                const device_vector<FieldT> coeffs = dev::IFFT<FieldT>(by_interaction[j][l], domains_[num_reductions_]);                                     // :538
                IOP_.submit_prover_message(final_polynomial_handles_[j][l], coeffs.to_host(final_polynomial_degree_bound_));
            }
        IOP_.signal_prover_round_done();
    }
};

template<typename FieldT>
class LDT_instance_reducer {                                                                  // ldt_reducer.tcc:134-297, non-zk, multi_LDT = FRI_protocol
    bcs_prover<FieldT> &IOP_;
    domain_handle codeword_domain_handle_;
    std::size_t num_output_LDT_instances_, max_tested_degree_bound_;
    std::vector<std::shared_ptr<combined_LDT_device_oracle<FieldT>>> combined_oracles_;
    std::vector<oracle_handle> combined_oracle_handles_;
    std::vector<verifier_random_message_handle> random_coefficients_handles_;
    std::shared_ptr<FRI_protocol<FieldT>> multi_LDT_;
public:
    LDT_instance_reducer(bcs_prover<FieldT> &IOP, const domain_handle &codeword_domain_handle, std::size_t num_output_LDT_instances, std::size_t max_tested_degree_bound)
        : IOP_(IOP), codeword_domain_handle_(codeword_domain_handle), num_output_LDT_instances_(num_output_LDT_instances), max_tested_degree_bound_(max_tested_degree_bound) {}
    void register_interactions(const std::vector<oracle_handle> &oracle_handles, const std::vector<std::size_t> &localization_parameters,
                               std::size_t fri_interactive_repetitions, std::size_t fri_query_repetitions)
    {
        std::vector<std::size_t> degrees;
        for (auto &h : oracle_handles) {
            degrees.push_back(IOP_.get_oracle_degree(h));
            if (degrees.back() > max_tested_degree_bound_)
                throw std::invalid_argument("One of the oracles is registered with claimed degree " + std::to_string(degrees.back()) +
                                            ", which is greater than the max tested degree bound");
        }
        const field_subset<FieldT> L = IOP_.get_domain(codeword_domain_handle_);
        for (std::size_t i = 0; i < num_output_LDT_instances_; ++i) combined_oracles_.push_back(std::make_shared<combined_LDT_device_oracle<FieldT>>(L, degrees));
        for (auto &o : combined_oracles_) combined_oracle_handles_.push_back(IOP_.register_virtual_oracle(codeword_domain_handle_, max_tested_degree_bound_, oracle_handles, o));
        for (std::size_t i = 0; i < num_output_LDT_instances_; ++i) random_coefficients_handles_.push_back(IOP_.register_verifier_random_message(2 * oracle_handles.size()));
        IOP_.set_head_hint((std::size_t)1 << detail::log2_ceil(max_tested_degree_bound_));
        multi_LDT_ = std::make_shared<FRI_protocol<FieldT>>(IOP_, codeword_domain_handle_, combined_oracle_handles_, localization_parameters, max_tested_degree_bound_,
                                                            fri_interactive_repetitions, fri_query_repetitions);
        multi_LDT_->register_interactions();
    }
    void register_queries() { multi_LDT_->register_queries(); }
    void calculate_and_submit_proof()                                                        // :259-272
    {
        for (std::size_t i = 0; i < combined_oracles_.size(); ++i)
            combined_oracles_[i]->set_random_coefficients(IOP_.obtain_verifier_random_message(random_coefficients_handles_[i]));
        multi_LDT_->calculate_and_submit_proof();
    }
};

template<typename FieldT>
class aurora_iop {                                                                            // aurora_iop.tcc:262-344
    bcs_prover<FieldT> &IOP_;
    const aurora_snark_parameters<FieldT> &params_;
    domain_handle codeword_domain_handle_;
    std::shared_ptr<encoded_aurora_protocol<FieldT>> protocol_;
    std::shared_ptr<LDT_instance_reducer<FieldT>> LDT_reducer_;
    field_subset<FieldT> quotient_map_domain_;
public:
    aurora_iop(bcs_prover<FieldT> &IOP, const r1cs_constraint_system<FieldT> &constraint_system, const aurora_snark_parameters<FieldT> &params)
        : IOP_(IOP), params_(params)
    {
        const FieldT codeword_domain_shift = field_subset<FieldT>((std::size_t)1 << params.codeword_domain_dim_).element_outside_of_subset();   // :282-283
        const domain_handle constraint_h = IOP.register_domain(field_subset<FieldT>((std::size_t)1 << params.constraint_domain_dim_));
        const domain_handle variable_h = IOP.register_domain(field_subset<FieldT>((std::size_t)1 << params.variable_domain_dim_));
        codeword_domain_handle_ = IOP.register_domain(dist::mark_codeword_domain(field_subset<FieldT>((std::size_t)1 << params.codeword_domain_dim_, codeword_domain_shift),
                                                                                 (std::size_t)1 << params.localization_parameters_[0]));
        protocol_ = std::make_shared<encoded_aurora_protocol<FieldT>>(IOP, constraint_h, variable_h, codeword_domain_handle_, constraint_system,
                                                                      params.multi_lincheck_repetitions_);
        LDT_reducer_ = std::make_shared<LDT_instance_reducer<FieldT>>(IOP, codeword_domain_handle_, params.num_output_LDT_instances_, params.max_tested_degree_bound_);
        quotient_map_domain_ = IOP.get_domain(codeword_domain_handle_).get_subset_of_order((std::size_t)1 << params.localization_parameters_[0]);
        IOP.set_round_parameters(quotient_map_domain_);                                      // :307-308
    }
    void register_interactions()                                                             // :311-326
    {
        protocol_->register_challenge();
        protocol_->register_proof();
        IOP_.set_round_parameters(quotient_map_domain_);
        LDT_reducer_->register_interactions(protocol_->get_all_oracle_handles(), params_.localization_parameters_, params_.fri_interactive_repetitions_,
                                            params_.fri_query_repetitions_);
    }
    void register_queries() { LDT_reducer_->register_queries(); }
    bool witness_submitted_ = false;
    // produce_proof in two halves: the witness oracles' kernels are enqueued by the first (no host synchronisation), so that a caller may finish host-side
    // set-up that the first round does not need — the query registrations — while the GPU works
    void submit_witness(const std::vector<FieldT> &primary_input, const std::vector<FieldT> &auxiliary_input, const device_vector<FieldT> *d_assignment = nullptr)
    {
        protocol_->submit_witness_oracles(primary_input, auxiliary_input, d_assignment);
        witness_submitted_ = true;
    }
    void produce_proof(const std::vector<FieldT> &primary_input, const std::vector<FieldT> &auxiliary_input, const device_vector<FieldT> *d_assignment = nullptr)   // :334-344
    {
        if (!witness_submitted_) protocol_->submit_witness_oracles(primary_input, auxiliary_input, d_assignment);
        IOP_.signal_prover_round_done();
        protocol_->calculate_and_submit_proof();
        IOP_.signal_prover_round_done();
        LDT_reducer_->calculate_and_submit_proof();
    }
};

namespace detail {
// registration, rounds, then `finish(IOP)`: the structured transcript or its canonical bytes
template<typename FieldT, typename Finish>
auto run_aurora_prover(const r1cs_constraint_system<FieldT> &constraint_system, const r1cs_primary_input<FieldT> &primary_input,
                       const r1cs_auxiliary_input<FieldT> &auxiliary_input, const aurora_snark_parameters<FieldT> &parameters,
                       const device_vector<FieldT> *d_assignment, Finish finish) -> decltype(finish(std::declval<bcs_prover<FieldT> &>()))
{
    // IOPX_HOST_TIMING=1: wall-clock marks of the host-side phases on stderr (registration before the first kernel, the rounds, transcript extraction)
    const bool timing = iopx_get_option("IOPX_HOST_TIMING", 0) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (timing) std::fprintf(stderr, "[iopx host] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    };
    bcs_prover<FieldT> IOP(parameters.pow_bits_);
    aurora_iop<FieldT> full_protocol(IOP, constraint_system, parameters);
    full_protocol.register_interactions();
    IOP.seal_interaction_registrations();
    mark("interactions registered");
    full_protocol.submit_witness(primary_input, auxiliary_input, d_assignment);     // round 1's kernels are in flight while the queries are registered
    mark("witness oracles enqueued");
    full_protocol.register_queries();
    IOP.seal_query_registrations();
    mark("queries registered");
    full_protocol.produce_proof(primary_input, auxiliary_input, d_assignment);
    mark("rounds done (enqueued)");
    auto result = finish(IOP);
    mark("transcript extracted");
    return result;
}
} // namespace detail

// aurora_snark_prover (aurora_snark.tcc:119-146).  With d_assignment — the (1, primary, auxiliary) vector already in HBM — the
// witness never crosses PCIe inside the call.
template<typename FieldT>
bcs_transformation_transcript<FieldT> aurora_snark_prover(const r1cs_constraint_system<FieldT> &constraint_system, const r1cs_primary_input<FieldT> &primary_input,
                                                          const r1cs_auxiliary_input<FieldT> &auxiliary_input, const aurora_snark_parameters<FieldT> &parameters,
                                                          const device_vector<FieldT> *d_assignment = nullptr)
{
    return detail::run_aurora_prover<FieldT>(constraint_system, primary_input, auxiliary_input, parameters, d_assignment,
                                             [](bcs_prover<FieldT> &IOP) { return IOP.get_transcript(); });
}

// ... returning aurora_snark_prover(...).serialize() without building the structured transcript (bcs_prover::get_transcript_bytes): what the C ABI hands out
template<typename FieldT>
std::string aurora_snark_prover_serialized(const r1cs_constraint_system<FieldT> &constraint_system, const r1cs_primary_input<FieldT> &primary_input,
                                           const r1cs_auxiliary_input<FieldT> &auxiliary_input, const aurora_snark_parameters<FieldT> &parameters,
                                           const device_vector<FieldT> *d_assignment = nullptr)
{
    return detail::run_aurora_prover<FieldT>(constraint_system, primary_input, auxiliary_input, parameters, d_assignment,
                                             [](bcs_prover<FieldT> &IOP) { return IOP.get_transcript_bytes(); });
}

} // namespace libiop_amd
