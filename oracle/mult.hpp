// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's multiplicative-domain path (prime fields): degree-aware radix-2 FFT,
// the libfqfft inverse transforms it wraps, IFFT of known degree and the multiplicative FRI fold.
// libfqfft (scipr-lab/libfqfft, absent submodule, unpinned) is restated from its published algorithm:
//   basic_radix2_domain::iFFT(a)        = radix-2 FFT with omega^-1 (CLRS bit-reversal + butterflies), then a[i] *= m^-1
//   basic_radix2_domain::icosetFFT(a,g) = iFFT(a) then a[i] *= g^-i
//   _multiply_by_coset(a, g)            : a[i] *= g^i for i >= 1
// with omega = the generator of the order-m subgroup (= libiop's subgroup generator).
// Citations are relative to /root/reference.
#pragma once
#include <cassert>
#include <vector>
#include "algebra.hpp"
#include "fp.hpp"

namespace oracle {

// multiplicative_coset: order 2^k, generator g = multiplicative_generator^((p-1)/order), shift
// (libiop/algebra/field_subset/subgroup.tcc:33-75, 199-260)
template<typename F>
struct mult_coset {
    size_t order;
    F g, shift;
    mult_coset(size_t n, const F &s = F::one()) : order(n), g(F::subgroup_generator(n)), shift(s) {}
    size_t dimension() const { return ceil_log2(order); }
    std::vector<F> all_elements() const       // shift * g^i
    {
        std::vector<F> out;
        F cur = shift;
        for (size_t i = 0; i < order; ++i) { out.push_back(cur); cur *= g; }
        return out;
    }
    // field_subset.tcc:217-237 (multiplicative): default subgroup of that order, same shift
    mult_coset subset_of_order(size_t o) const { return mult_coset(o, shift); }
};

// subgroup.tcc:117-144 — levels concatenated: level s (m = 2^(s-1)) at offset m - 1, entry j = (g^(order/2m))^j
template<typename F>
std::vector<F> fft_cache(const mult_coset<F> &c)
{
    std::vector<F> elems;
    size_t m = 1;
    for (size_t s = 1; s <= c.dimension(); ++s) {
        const F w_m = c.g.pow(c.order / (2 * m));
        F w = F::one();
        for (size_t j = 0; j < m; ++j) { elems.push_back(w); w *= w_m; }
        m *= 2;
    }
    return elems;
}

// libfqfft _multiply_by_coset
template<typename F>
void multiply_by_coset(std::vector<F> &a, const F &g)
{
    F u = g;
    for (size_t i = 1; i < a.size(); ++i) { a[i] *= u; u *= g; }
}

// libiop/algebra/fft.tcc:236-317
template<typename F>
std::vector<F> multiplicative_FFT_degree_aware(const std::vector<F> &poly_coeffs, const mult_coset<F> &coset)
{
    assert(poly_coeffs.size() <= coset.order);
    const size_t n = coset.order, logn = ceil_log2(n);
    std::vector<F> a(poly_coeffs);
    if (coset.shift != F::one()) multiply_by_coset<F>(a, coset.shift);          // :246-249
    a.resize(n, F::zero());
    const size_t poly_dimension = ceil_log2(poly_coeffs.size());                 // :252 (ceil log)
    const size_t poly_size = poly_coeffs.size();
    const size_t dup = (size_t)1 << (logn - poly_dimension);                     // :263
    for (size_t k = 0; k < poly_size; ++k) {                                     // :267-274
        const size_t rk = bitreverse(k, logn);
        if (k < rk) std::swap(a[k], a[rk]);
    }
    if (dup > 1) {                                                               // :280-289
        for (size_t i = 0; i < n; i += dup)
            for (size_t j = 1; j < dup; ++j) a[i + j] = a[i];
    }
    const std::vector<F> cache = fft_cache<F>(coset);
    size_t m = (size_t)1 << (logn - poly_dimension);
    for (size_t s = logn - poly_dimension + 1; s <= logn; ++s) {                 // :293-315
        const size_t w_index_base = m - 1;
        for (size_t k = 0; k < n; k += 2 * m) {
            for (size_t j = 0; j < m; ++j) {
                const F t = cache[w_index_base + j] * a[k + j + m];
                a[k + j + m] = a[k + j] - t;
                a[k + j] += t;
            }
        }
        m *= 2;
    }
    return a;
}

// libfqfft _basic_serial_radix2_FFT (CLRS 2nd ed. p. 864)
template<typename F>
void basic_radix2_FFT(std::vector<F> &a, const F &omega)
{
    const size_t n = a.size(), logn = ceil_log2(n);
    assert(n == ((size_t)1 << logn));
    for (size_t k = 0; k < n; ++k) {
        const size_t rk = bitreverse(k, logn);
        if (k < rk) std::swap(a[k], a[rk]);
    }
    size_t m = 1;
    for (size_t s = 1; s <= logn; ++s) {
        const F w_m = omega.pow(n / (2 * m));
        for (size_t k = 0; k < n; k += 2 * m) {
            F w = F::one();
            for (size_t j = 0; j < m; ++j) {
                const F t = w * a[k + j + m];
                a[k + j + m] = a[k + j] - t;
                a[k + j] += t;
                w *= w_m;
            }
        }
        m *= 2;
    }
}

// libiop/algebra/fft.tcc:343-361 over libfqfft's iFFT / icosetFFT; :397-401 size-1 early return
template<typename F>
std::vector<F> multiplicative_IFFT(const std::vector<F> &evals, const mult_coset<F> &coset)
{
    assert(evals.size() == coset.order);
    if (evals.size() == 1) return evals;
    std::vector<F> vec(evals);
    basic_radix2_FFT<F>(vec, coset.g.inverse());
    const F sconst = F((uint64_t)coset.order).inverse();
    for (F &v : vec) v *= sconst;
    if (coset.shift != F::one()) multiply_by_coset<F>(vec, coset.shift.inverse());
    return vec;
}

// libiop/algebra/fft.tcc:435-456 — every (n / 2^ceil(log2 degree))-th evaluation, IFFT over that sub-coset
template<typename F>
std::vector<F> multiplicative_IFFT_of_known_degree(const std::vector<F> &evals, size_t degree, const mult_coset<F> &coset)
{
    const size_t pow2 = (size_t)1 << ceil_log2(degree);
    const mult_coset<F> minimal = coset.subset_of_order(pow2);
    std::vector<F> sub;
    const size_t freq = coset.order / pow2;
    for (size_t i = 0; i < coset.order; i += freq) sub.push_back(evals[i]);
    return multiplicative_IFFT<F>(sub, minimal);
}

// libiop/protocols/ldt/fri/fri_aux.tcc:106-249 — cosets {j + k * n/c}; one global batch inversion
template<typename F>
std::vector<F> multiplicative_evaluate_next_f_i_over_entire_domain(const std::vector<F> &f_i_evals, const mult_coset<F> &dom,
                                                                   size_t coset_size, const F &x_i)
{
    const size_t num_cosets = dom.order / coset_size;
    std::vector<F> next;
    next.reserve(num_cosets);
    const F h_inc = dom.g;                                                               // :153
    const F h_inc_to_coset_inv_plus_one = h_inc.pow(coset_size).inverse() * h_inc;       // :154-155
    const F g = F::subgroup_generator(coset_size);                                       // :156-157
    const F g_inv = g.inverse();
    const F x_to_order_coset = x_i.pow(coset_size);
    std::vector<F> shifted_x(coset_size);
    shifted_x[0] = x_i;
    for (size_t i = 1; i < coset_size; ++i) shifted_x[i] = shifted_x[i - 1] * g_inv;     // :161-166
    F cur_h = dom.shift;
    const F first = cur_h.pow(coset_size).inverse() * cur_h;
    F cur_const_plus_h = x_to_order_coset * first;
    std::vector<F> to_invert;
    to_invert.reserve(dom.order);
    std::vector<F> coset_consts;
    const F const_all = F((uint64_t)coset_size).inverse();
    bool x_ever = false;
    size_t x_coset = 0, x_index = 0;
    for (size_t j = 0; j < num_cosets; ++j) {
        const F coset_constant = cur_const_plus_h - cur_h;                               // :192
        coset_consts.push_back(coset_constant);
        if (coset_constant == F::zero()) {                                               // :197-217 (incl. the `continue`, F9)
            x_ever = true; x_coset = j;
            F cur_elem = cur_h;
            for (size_t k = 0; k < coset_size; ++k) {
                if (cur_elem == x_i) x_index = k * num_cosets + j;
                cur_elem *= g;
                to_invert.push_back(F::one());
            }
            continue;
        }
        for (size_t k = 0; k < coset_size; ++k) to_invert.push_back(shifted_x[k] - cur_h); // :219-222
        cur_h *= h_inc;
        cur_const_plus_h *= h_inc_to_coset_inv_plus_one;
    }
    const std::vector<F> lagrange = batch_inverse_and_mul<F>(to_invert, const_all);       // :230-231
    for (size_t j = 0; j < num_cosets; ++j) {
        F interp = F::zero();
        for (size_t k = 0; k < coset_size; ++k) interp += f_i_evals[k * num_cosets + j] * lagrange[j * coset_size + k];
        interp *= coset_consts[j];
        next.push_back(interp);
    }
    if (x_ever) next[x_coset] = f_i_evals[x_index];
    return next;
}

// libiop/protocols/ldt/fri/fri_aux.tcc:305-349 — multiplicative_evaluate_next_f_i_at_coset: the verifier's single-coset fold.
//   g : generator of the order-|coset| subgroup, h : the coset's first element (shift of the queried coset)
template<typename F>
F multiplicative_evaluate_next_f_i_at_coset(const std::vector<F> &f_i_evals_over_coset, const F &g, const F &h, const F &x_i)
{
    const size_t coset_size = f_i_evals_over_coset.size();
    const F vp_x = x_i.pow(coset_size) - h.pow(coset_size);                                           // :317-318
    const bool x_in_domain = (vp_x == F::zero());
    const F c = vp_x * (F((uint64_t)coset_size) * h.pow(coset_size - 1)).inverse();                   // :320
    std::vector<F> shifted;
    F cur = h;
    for (size_t k = 0; k < coset_size; ++k) {
        if (x_in_domain && cur == x_i) return f_i_evals_over_coset[k];                                // :332-334
        shifted.push_back(x_i - cur);
        cur *= g;
    }
    const std::vector<F> inverted = batch_inverse_and_mul<F>(shifted, c);                             // :339
    F interpolation = F::zero(), unshifted = F::one();
    for (size_t k = 0; k < coset_size; ++k) {
        interpolation += inverted[k] * unshifted * f_i_evals_over_coset[k];                           // :342-347
        unshifted *= g;
    }
    return interpolation;
}

} // namespace oracle
