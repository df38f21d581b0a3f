// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's additive-FFT path, loop for loop (same sweeps, same operation
// order) so that it can serve both as the bit-exact checker and as the "port" CPU baseline.
// Each function cites the reference lines it follows (paths relative to /root/reference).
#pragma once
#include <cassert>
#include <cstdint>
#include <cstddef>
#include <stdexcept>
#include <utility>
#include <vector>

namespace oracle {

// libff::log2 = ceil(log2(n)) (libff/common/utils; used at libiop/algebra/utils.tcc:148, fft.tcc:242,252)
static inline size_t ceil_log2(size_t n)
{
    size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) { n >>= 1; ++r; }
    return r;
}

// libff::bitreverse(n, l): reverse the low l bits (used at libiop/algebra/utils.tcc:152, fft.tcc:269)
static inline size_t bitreverse(size_t n, size_t l)
{
    size_t r = 0;
    for (size_t k = 0; k < l; ++k) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}

// libiop/algebra/utils.tcc:8-30 — result[0] = shift, then doubling with basis[i]
template<typename T>
std::vector<T> all_subset_sums(const std::vector<T> &basis, const T &shift)
{
    std::vector<T> out;
    out.reserve((size_t)1 << basis.size());
    out.push_back(shift);
    for (size_t i = 0; i < basis.size(); ++i) {
        const size_t have = (size_t)1 << i;
        for (size_t j = 0; j < have; ++j) out.push_back(out[j] + basis[i]);
    }
    return out;
}

// libiop/algebra/utils.tcc:144-159
template<typename T>
void bitreverse_vector(std::vector<T> &a)
{
    const size_t n = a.size(), logn = ceil_log2(n);
    assert(n == ((size_t)1 << logn));
    for (size_t k = 0; k < n; ++k) {
        const size_t rk = bitreverse(k, logn);
        if (k < rk) std::swap(a[k], a[rk]);
    }
}

// libiop/algebra/utils.tcc:38-69 — Montgomery's trick, every output additionally scaled by k
template<typename F>
std::vector<F> batch_inverse_and_mul(const std::vector<F> &vec, const F &k)
{
    std::vector<F> R;
    R.reserve(vec.size());
    F c = vec[0];
    R.push_back(c);
    for (size_t i = 1; i < vec.size(); ++i) { c *= vec[i]; R.push_back(c); }
    F c_inv = c.inverse() * k;
    for (size_t i = vec.size() - 1; i > 0; --i) { R[i] = R[i - 1] * c_inv; c_inv *= vec[i]; }
    R[0] = c_inv;
    return R;
}

// Affine subspace: basis + shift (libiop/algebra/field_subset/subspace.tcc:47-108, 219-272)
template<typename F>
struct affine_subspace {
    std::vector<F> basis;
    F shift;

    affine_subspace() {}
    affine_subspace(const std::vector<F> &b, const F &s) : basis(b), shift(s) {}

    size_t dimension() const { return basis.size(); }
    size_t num_elements() const { return (size_t)1 << basis.size(); }
    std::vector<F> all_elements() const { return all_subset_sums<F>(basis, shift); }

    // subspace.tcc:93-108 — basis element i is FieldT(1ull << i)
    static affine_subspace standard(size_t dim, const F &s = F::zero())
    {
        std::vector<F> b;
        for (size_t i = 0; i < dim; ++i) b.push_back(F((uint64_t)1 << i));
        return affine_subspace(b, s);
    }

    // subspace.tcc:56-71 (+ shift, :250-255)
    F element_by_index(size_t idx) const
    {
        F r = shift;
        for (size_t i = 0; i < basis.size(); ++i) if (idx & ((size_t)1 << i)) r += basis[i];
        return r;
    }

    // field_subset.tcc:217-237 (additive): first log2(order) basis vectors, same shift
    affine_subspace subset_of_order(size_t order) const
    {
        const size_t d = ceil_log2(order);
        return affine_subspace(std::vector<F>(basis.begin(), basis.begin() + d), shift);
    }
};

// libiop/algebra/fft.tcc:12-37 — Horner evaluation at every element of the domain
template<typename F>
std::vector<F> naive_FFT(const std::vector<F> &coeffs, const std::vector<F> &points)
{
    std::vector<F> out;
    out.reserve(points.size());
    for (const F &p : points) {
        F v = F::zero();
        for (size_t i = coeffs.size(); i--; ) { v *= p; v += coeffs[i]; }
        out.push_back(v);
    }
    return out;
}

// libiop/algebra/fft.tcc:39-124 (Gao–Mateer additive FFT, iterative, in place)
template<typename F>
std::vector<F> additive_FFT(const std::vector<F> &poly_coeffs, const affine_subspace<F> &domain)
{
    std::vector<F> S(poly_coeffs);
    S.resize(domain.num_elements(), F::zero());                 // :42-44 zero padding
    const size_t n = S.size(), m = domain.dimension();
    assert(n == ((size_t)1 << m));

    std::vector<F> recursed_betas((m + 1) * m / 2, F::zero());
    std::vector<F> recursed_shifts(m, F::zero());
    size_t betas_ptr = 0;

    std::vector<F> betas2(domain.basis);
    F shift2 = domain.shift;
    for (size_t j = 0; j < m; ++j) {
        const F beta = betas2[m - 1 - j];
        F betai = F::one();
        // :62-70 twist: block (ofs >> j) is scaled by beta^(ofs >> j)
        for (size_t ofs = 0; ofs < n; ofs += ((size_t)1 << j)) {
            for (size_t p = 0; p < ((size_t)1 << j); ++p) S[ofs + p] *= betai;
            betai *= beta;
        }
        // :73-83 radix conversion (Taylor expansion at x^2 - x)
        for (size_t stride = n / 4; stride >= ((size_t)1 << j); stride >>= 1) {
            for (size_t ofs = 0; ofs < n; ofs += stride * 4) {
                for (size_t i = 0; i < stride; ++i) {
                    S[ofs + 2 * stride + i] += S[ofs + 3 * stride + i];
                    S[ofs + 1 * stride + i] += S[ofs + 2 * stride + i];
                }
            }
            if (stride == 0) break;
        }
        // :86-96 recursed basis / shift
        const F betainv = beta.inverse();
        for (size_t i = 0; i < m - 1 - j; ++i) {
            const F newbeta = betas2[i] * betainv;
            recursed_betas[betas_ptr++] = newbeta;
            betas2[i] = newbeta.squared() - newbeta;
        }
        const F newshift = shift2 * betainv;
        recursed_shifts[j] = newshift;
        shift2 = newshift.squared() - newshift;
    }

    bitreverse_vector<F>(S);                                     // :99

    // :102-120 unwind the recursion
    for (size_t j = 0; j < m; ++j) {
        betas_ptr -= j;
        const std::vector<F> popped(recursed_betas.begin() + betas_ptr, recursed_betas.begin() + betas_ptr + j);
        const F popped_shift = recursed_shifts[m - 1 - j];
        const std::vector<F> sums = all_subset_sums<F>(popped, popped_shift);
        const size_t stride = (size_t)1 << j;
        for (size_t ofs = 0; ofs < n; ofs += 2 * stride) {
            for (size_t i = 0; i < stride; ++i) {
                S[ofs + i] += S[ofs + stride + i] * sums[i];
                S[ofs + stride + i] += S[ofs + i];
            }
        }
    }
    assert(betas_ptr == 0);
    return S;
}

// libiop/algebra/fft.tcc:126-204
template<typename F>
std::vector<F> additive_IFFT(const std::vector<F> &evals, const affine_subspace<F> &domain)
{
    const size_t n = evals.size(), m = domain.dimension();
    assert(n == ((size_t)1 << m));
    std::vector<F> S(evals);
    std::vector<F> recursed_twists(m, F::zero());

    std::vector<F> betas2(domain.basis);
    F shift2 = domain.shift;
    for (size_t j = 0; j < m; ++j) {
        const F beta = betas2[m - 1 - j];
        const F betainv = beta.inverse();
        recursed_twists[j] = betainv;
        std::vector<F> newbetas(m - 1 - j, F::zero());
        for (size_t i = 0; i < m - 1 - j; ++i) {
            const F nb = betas2[i] * betainv;
            newbetas[i] = nb;
            betas2[i] = nb.squared() - nb;
        }
        const F newshift = shift2 * betainv;
        shift2 = newshift.squared() - newshift;
        const std::vector<F> sums = all_subset_sums<F>(newbetas, newshift);
        const size_t half = (size_t)1 << (m - 1 - j);
        for (size_t ofs = 0; ofs < n; ofs += 2 * half) {        // :160-167
            for (size_t p = 0; p < half; ++p) {
                S[ofs + half + p] += S[ofs + p];
                S[ofs + p] += S[ofs + half + p] * sums[p];
            }
        }
    }

    bitreverse_vector<F>(S);                                     // :170

    for (size_t j = 0; j < m; ++j) {
        size_t N = (size_t)4 << (m - 1 - j);
        while (N <= n) {                                         // :175-188 radix combinations
            const size_t quarter = N / 4;
            for (size_t ofs = 0; ofs < n; ofs += N) {
                for (size_t i = 0; i < quarter; ++i) {
                    S[ofs + 1 * quarter + i] += S[ofs + 2 * quarter + i];
                    S[ofs + 2 * quarter + i] += S[ofs + 3 * quarter + i];
                }
            }
            N *= 2;
        }
        const F betainv = recursed_twists[m - 1 - j];            // :190-200 untwist
        F betainvi = F::one();
        const size_t blk = (size_t)1 << (m - 1 - j);
        for (size_t ofs = 0; ofs < n; ofs += blk) {
            for (size_t p = 0; p < blk; ++p) S[ofs + p] *= betainvi;
            betainvi *= betainv;
        }
    }
    return S;
}

// libiop/algebra/fft.tcc:458-475 (additive): IFFT over the first 2^ceil(log2 degree) evaluations
template<typename F>
std::vector<F> additive_IFFT_of_known_degree(const std::vector<F> &evals, size_t degree, const affine_subspace<F> &domain)
{
    const size_t pow2 = (size_t)1 << ceil_log2(degree);
    const affine_subspace<F> minimal = domain.subset_of_order(pow2);
    const std::vector<F> head(evals.begin(), evals.begin() + pow2);
    return additive_IFFT<F>(head, minimal);
}

// ---- linearized polynomials (only what the FRI fold / domain chain needs) -----------------------
// coefficient i multiplies X^(2^(i-1)) for i >= 1; coefficient 0 is the constant term
// (libiop/algebra/polynomials/linearized_polynomial.tcc:29-48)
template<typename F>
F linearized_eval(const std::vector<F> &c, const F &x)
{
    if (c.empty()) return F::zero();
    F r = c[0];
    F xp = x;
    for (size_t i = 1; i < c.size(); ++i) { r += c[i] * xp; xp = xp.squared(); }
    return r;
}

// libiop/algebra/polynomials/vanishing_polynomial.tcc:373-395 — Z_{<S>}(y), built one basis vector at
// a time: Z_k = Z_{k-1}^2 + Z_{k-1}(s_k) * Z_{k-1}; finally constant term += Z(shift).
template<typename F>
std::vector<F> vanishing_polynomial_from_subspace(const affine_subspace<F> &S)
{
    std::vector<F> poly = { F::zero(), F::one() };
    for (const F &c : S.basis) {
        const F pc = linearized_eval<F>(poly, c);
        // squared(): every coefficient (slot 0 included) moves up one slot and is squared, slot 0
        // becomes 0 (linearized_polynomial.tcc:83-100); slot 0 is zero throughout this loop
        std::vector<F> sq(poly.size() + 1, F::zero());
        for (size_t i = poly.size(); i > 0; --i) sq[i] = poly[i - 1].squared();
        for (size_t i = 0; i < poly.size(); ++i) sq[i] += poly[i] * pc;
        poly = sq;
    }
    poly[0] += linearized_eval<F>(poly, S.shift);
    return poly;
}

} // namespace oracle
