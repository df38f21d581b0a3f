// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// One vocabulary over the two evaluation-domain kinds, as the reference's tagged union field_subset<FieldT> offers
// (libiop/algebra/field_subset/field_subset.tcc), plus the vanishing-polynomial helpers the encoded protocols use
// (libiop/algebra/polynomials/vanishing_polynomial.tcc, linearized_polynomial.tcc).  A field has exactly one domain kind
// (libff::is_additive / is_multiplicative), so the union is resolved at compile time: domain_of<F>.
// Citations are relative to /root/reference.
#pragma once
#include <type_traits>
#include <utility>
#include <vector>
#include "algebra.hpp"
#include "field.hpp"
#include "fp.hpp"
#include "fri.hpp"
#include "mult.hpp"

namespace oracle {

template<typename F> struct field_info;
template<int W, uint64_t T> struct field_info<gf2n<W, T>> {
    static constexpr bool additive = true;
    static size_t soundness_log_of_field_size() { return 64 * W; }                  // extension degree (libff, recalled)
};
template<typename P> struct field_info<Fp<P>> {
    static constexpr bool additive = false;
    static size_t soundness_log_of_field_size()                                    // floor(log2 p) (libff, recalled): 180 for edwards_Fr
    {
        size_t bits = 0;
        for (int i = P::limbs - 1; i >= 0 && bits == 0; --i)
            for (int b = 63; b >= 0; --b) if ((P::modulus[i] >> b) & 1) { bits = 64 * i + b + 1; break; }
        return bits - 1;
    }
};

template<typename F>
using domain_of = typename std::conditional<field_info<F>::additive, affine_subspace<F>, mult_coset<F>>::type;

// ---- construction (field_subset.tcc:3-62) ----
template<typename F> affine_subspace<F> make_domain(size_t num_elements, const F &shift, const affine_subspace<F> *)
{
    return affine_subspace<F>::standard(ceil_log2(num_elements), shift);
}
template<typename F> mult_coset<F> make_domain(size_t num_elements, const F &shift, const mult_coset<F> *)
{
    return mult_coset<F>(num_elements, shift);
}
template<typename F> domain_of<F> default_domain(size_t num_elements)                    // :3-11
{
    return make_domain<F>(num_elements, field_info<F>::additive ? F::zero() : F::one(), (const domain_of<F> *)nullptr);
}
template<typename F> domain_of<F> shifted_domain(size_t num_elements, const F &shift)    // :13-18
{
    return make_domain<F>(num_elements, shift, (const domain_of<F> *)nullptr);
}

// ---- queries ----
template<typename F> bool dom_additive(const affine_subspace<F> &) { return true; }
template<typename F> bool dom_additive(const mult_coset<F> &) { return false; }
template<typename F> size_t dom_size(const affine_subspace<F> &d) { return d.num_elements(); }
template<typename F> size_t dom_size(const mult_coset<F> &d) { return d.order; }
template<typename F> size_t dom_dim(const affine_subspace<F> &d) { return d.dimension(); }
template<typename F> size_t dom_dim(const mult_coset<F> &d) { return d.dimension(); }
template<typename F> F dom_shift(const affine_subspace<F> &d) { return d.shift; }
template<typename F> F dom_shift(const mult_coset<F> &d) { return d.shift; }
template<typename F> F dom_element(const affine_subspace<F> &d, size_t i) { return d.element_by_index(i); }
template<typename F> F dom_element(const mult_coset<F> &d, size_t i) { return d.shift * d.g.pow(i); }
template<typename F> std::vector<F> dom_elements(const affine_subspace<F> &d) { return d.all_elements(); }
template<typename F> std::vector<F> dom_elements(const mult_coset<F> &d) { return d.all_elements(); }
template<typename F> affine_subspace<F> dom_subset_of_order(const affine_subspace<F> &d, size_t o) { return d.subset_of_order(o); }
template<typename F> mult_coset<F> dom_subset_of_order(const mult_coset<F> &d, size_t o) { return d.subset_of_order(o); }
// element_outside_of_subset: subspace.tcc:219-227 (standard basis), subgroup.tcc:311-315
template<typename F> F dom_element_outside(const affine_subspace<F> &d) { return d.shift + F((uint64_t)1 << d.dimension()); }
template<typename F> F dom_element_outside(const mult_coset<F> &d) { return d.shift * F::multiplicative_generator(); }

// reindex_by_subset (field_subset.tcc:130-142): identity for subspaces; subgroup.tcc:149-173 for cosets
template<typename F> size_t dom_reindex_by_subset(const affine_subspace<F> &, size_t, size_t index) { return index; }
template<typename F> size_t dom_reindex_by_subset(const mult_coset<F> &d, size_t reindex_dim, size_t index)
{
    const size_t order_s = (size_t)1 << reindex_dim, order_g_over_s = (size_t)1 << (d.dimension() - reindex_dim);
    if (index < order_s) return index * order_g_over_s;
    const size_t i = index - order_s, x = order_g_over_s - 1;
    return i + (i / x) + 1;
}

// coset index maps (subspace.tcc:73-91, subgroup.tcc:175-197)
template<typename D> size_t dom_coset_index(const D &d, size_t pos, size_t cs) { return coset_index(dom_additive(d), dom_size(d), pos, cs); }
template<typename D> size_t dom_intra_coset_index(const D &d, size_t pos, size_t cs) { return intra_coset_index(dom_additive(d), dom_size(d), pos, cs); }
template<typename D> size_t dom_position(const D &d, size_t cidx, size_t intra, size_t cs) { return position_by_coset_indices(dom_additive(d), dom_size(d), cidx, intra, cs); }

// ---- transforms (fft.tcc:407-475) ----
// the timed blocks carry the names of the reference's printing wrappers (fft.tcc:206-228, 378-405), which every prover-side transform goes through
template<typename F> std::vector<F> FFT_over(const std::vector<F> &c, const affine_subspace<F> &d) { timed_block tb("Call to additive_FFT_wrapper"); return additive_FFT<F>(c, d); }
template<typename F> std::vector<F> FFT_over(const std::vector<F> &c, const mult_coset<F> &d) { timed_block tb("Call to multiplicative_FFT_wrapper"); return multiplicative_FFT_degree_aware<F>(c, d); }
template<typename F> std::vector<F> IFFT_over(const std::vector<F> &e, const affine_subspace<F> &d) { timed_block tb("Call to additive_IFFT_wrapper"); return additive_IFFT<F>(e, d); }
template<typename F> std::vector<F> IFFT_over(const std::vector<F> &e, const mult_coset<F> &d) { timed_block tb("Call to multiplicative_IFFT_wrapper"); return multiplicative_IFFT<F>(e, d); }
template<typename F> std::vector<F> IFFT_of_known_degree_over(const std::vector<F> &e, size_t deg, const affine_subspace<F> &d)
{
    timed_block tb("Call to additive_IFFT_wrapper");
    return additive_IFFT_of_known_degree<F>(e, deg, d);
}
template<typename F> std::vector<F> IFFT_of_known_degree_over(const std::vector<F> &e, size_t deg, const mult_coset<F> &d)
{
    timed_block tb("Call to multiplicative_IFFT_wrapper");
    return multiplicative_IFFT_of_known_degree<F>(e, deg, d);
}

// ---- vanishing polynomial of a domain (vanishing_polynomial.tcc:14-120) ----
template<typename F, typename D> struct vanishing_polynomial;

template<typename F>
struct vanishing_polynomial<F, affine_subspace<F>> {
    std::vector<F> lin;                       // linearized: slot 0 constant, slot i >= 1 multiplies X^(2^(i-1))
    size_t degree;
    explicit vanishing_polynomial(const affine_subspace<F> &S) : lin(vanishing_polynomial_from_subspace<F>(S)), degree(S.num_elements()) {}
    F evaluation_at_point(const F &x) const { return linearized_eval<F>(lin, x); }
    std::vector<F> evaluations_over(const affine_subspace<F> &S) const
    {
        std::vector<F> out;
        for (const F &x : S.all_elements()) out.push_back(evaluation_at_point(x));
        return out;
    }
    // polynomial_over_linearized_polynomial (linearized_polynomial.tcc:238-289): (quotient, remainder)
    std::pair<std::vector<F>, std::vector<F>> divide(const std::vector<F> &P) const
    {
        const F linv = lin.back().inverse();
        const size_t deg_Z = degree;
        if (P.empty() || P.size() - 1 < deg_Z) { std::vector<F> r(P); r.resize(deg_Z, F::zero()); return { std::vector<F>(), r }; }
        std::vector<F> quotient(P.begin() + deg_Z, P.end()), remainder(P.begin(), P.begin() + deg_Z);
        for (size_t i = quotient.size(); i--; ) {
            const F twist = quotient[i] * linv;
            quotient[i] = twist;
            if (lin.size() >= 2) {
                size_t p_pow = deg_Z / 2;
                for (size_t j = lin.size() - 1; j--; ) {
                    if (i + p_pow < deg_Z) remainder[i + p_pow] -= twist * lin[j];
                    else quotient[i + p_pow - deg_Z] -= twist * lin[j];
                    p_pow /= 2;
                }
            }
        }
        return { quotient, remainder };
    }
};

template<typename F>
struct vanishing_polynomial<F, mult_coset<F>> {
    F vp_shift;                               // Z(X) = X^degree - shift^degree
    size_t degree;
    explicit vanishing_polynomial(const mult_coset<F> &S) : vp_shift(S.shift.pow(S.order)), degree(S.order) {}
    F evaluation_at_point(const F &x) const { return x.pow(degree) - vp_shift; }
    std::vector<F> evaluations_over(const mult_coset<F> &S) const
    {
        std::vector<F> out;
        for (const F &x : S.all_elements()) out.push_back(evaluation_at_point(x));
        return out;
    }
    // polynomial_over_multiplicative_vanishing_polynomial (vanishing_polynomial.tcc:314-358); its Z_0 is the stored
    // constant_coefficient() = -shift^degree ... the reference passes Z.constant_coefficient() as `vp_shift` and SUBTRACTS
    // twist * Z_0, i.e. it clears the term with Z's real constant coefficient.
    std::pair<std::vector<F>, std::vector<F>> divide(const std::vector<F> &P) const
    {
        const F Z_0 = -vp_shift;
        if (P.empty() || P.size() - 1 < degree) { std::vector<F> r(P); r.resize(degree, F::zero()); return { std::vector<F>(), r }; }
        std::vector<F> quotient(P.begin() + degree, P.end()), remainder(P.begin(), P.begin() + degree);
        for (size_t i = quotient.size(); i--; ) {
            const F twist = quotient[i];
            if (i < degree) remainder[i] -= twist * Z_0;
            else quotient[i - degree] -= twist * Z_0;
        }
        return { quotient, remainder };
    }
};

template<typename F> F poly_eval(const std::vector<F> &c, const F &x)        // polynomial.tcc:103-114 (Horner)
{
    F r = F::zero();
    for (size_t i = c.size(); i--; ) { r *= x; r += c[i]; }
    return r;
}

} // namespace oracle
