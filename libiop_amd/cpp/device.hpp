// Host scalars of FieldT and device-resident storage for C++ callers of the C ABI: field_host<FieldT> (per-proof constants),
// device_array<T> / device_vector<FieldT> (pooled HBM blocks, iopx_pool_alloc) — the storage behind oracle<FieldT>, where the reference
// has std::vector<FieldT> on the heap (libiop/iop/oracles.hpp:22-52).
#pragma once
#include <algorithm>
#include <functional>
#include <map>
#include <set>

#include "libiop_amd.hpp"

namespace libiop_amd {

// ---- host scalars of FieldT (per-proof constants only) --------------------------------------------------------------------------
template<typename FieldT>
struct field_host {
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates 24-byte field elements (libff::gf192 / libff::edwards_Fr layout)");
    static bool additive() { return field_kind<FieldT>::type == affine_subspace_type; }
    static FieldT from_words(const uint64_t *w) { FieldT r; std::memcpy((void *)&r, w, 24); return r; }
    static FieldT zero() { const uint64_t w[3] = { 0, 0, 0 }; return from_words(w); }
    static FieldT from_uint(uint64_t v)
    {
        uint64_t w[3] = { v, 0, 0 };
        if (!additive()) check(iopx_fp3_from_uint(v, w));
        return from_words(w);
    }
    static FieldT one() { return from_uint(1); }
    static bool is_zero(const FieldT &a) { const uint64_t *w = detail::words(&a); return (w[0] | w[1] | w[2]) == 0; }
    static FieldT add(const FieldT &a, const FieldT &b)
    {
        uint64_t w[3];
        if (additive()) for (int i = 0; i < 3; ++i) w[i] = detail::words(&a)[i] ^ detail::words(&b)[i];
        else check(iopx_fp3_host_add(detail::words(&a), detail::words(&b), w));
        return from_words(w);
    }
    static FieldT mul(const FieldT &a, const FieldT &b)
    {
        uint64_t w[3];
        if (additive()) check(iopx_gf192_host_mul(detail::words(&a), detail::words(&b), w));
        else check(iopx_fp3_host_mul(detail::words(&a), detail::words(&b), w));
        return from_words(w);
    }
    static FieldT pow(const FieldT &a, uint64_t e)
    {
        if (!additive()) { uint64_t w[3]; check(iopx_fp3_host_pow(detail::words(&a), e, w)); return from_words(w); }
        FieldT r = one(), b = a;
        for (; e; e >>= 1) { if (e & 1) r = mul(r, b); b = mul(b, b); }
        return r;
    }
    static FieldT sub(const FieldT &a, const FieldT &b)
    {
        if (additive()) return add(a, b);
        uint64_t w[3];
        check(iopx_fp3_host_sub(detail::words(&a), detail::words(&b), w));
        return from_words(w);
    }
    static FieldT neg(const FieldT &a) { return sub(zero(), a); }
    static FieldT inverse(const FieldT &a)
    {
        uint64_t w[3];
        if (additive()) check(iopx_gf192_inverse_host(detail::words(&a), w));
        else check(iopx_fp3_host_inverse(detail::words(&a), w));
        return from_words(w);
    }
    // Z_S(x) for the domain S (vanishing_polynomial::evaluation_at_point): the linearized polynomial of the subspace through the
    // library's host helper / x^|S| - shift^|S| (vanishing_polynomial.tcc:14-25)
    static FieldT vanishing_eval(const field_subset<FieldT> &S, const FieldT &x)
    {
        if (S.type() == affine_subspace_type) {
            uint64_t w[3];
            const FieldT shift = S.shift();
            check(iopx_gf192_vanishing_host(detail::words(S.basis().data()), S.dimension(), detail::words(&shift), detail::words(&x), w, nullptr));
            return from_words(w);
        }
        return sub(pow(x, S.num_elements()), pow(S.shift(), S.num_elements()));
    }
    // (DZ_S)(x): the linear coefficient for subspaces (vanishing_polynomial.tcc:63-72), |S| x^(|S| - 1) for cosets (:57-62)
    static FieldT vanishing_derivative(const field_subset<FieldT> &S, const FieldT &x)
    {
        if (S.type() == affine_subspace_type) {
            uint64_t w[3];
            const FieldT shift = S.shift();
            check(iopx_gf192_vanishing_host(detail::words(S.basis().data()), S.dimension(), detail::words(&shift), detail::words(&x), nullptr, w));
            return from_words(w);
        }
        return mul(from_uint(S.num_elements()), pow(x, S.num_elements() - 1));
    }
    // membership of x in S: standard-basis subspaces (x + shift below 2^dim) / (x / shift)^|S| = 1
    static bool element_in_domain(const field_subset<FieldT> &S, const FieldT &x)
    {
        if (S.type() == affine_subspace_type) {
            if (!S.subspace().is_standard_basis()) throw std::logic_error("membership test for a non-standard basis");
            const FieldT v = add(x, S.shift());
            const uint64_t *w = detail::words(&v);
            return w[1] == 0 && w[2] == 0 && (S.dimension() >= 64 || w[0] < ((uint64_t)1 << S.dimension()));
        }
        const FieldT r = pow(mul(x, inverse(S.shift())), S.num_elements());
        const FieldT o = one();
        return std::memcmp(&r, &o, sizeof(FieldT)) == 0;
    }
    // libff::soundness_log_of_field_size_helper: the extension degree for binary fields, floor(log2 p) for prime fields
    static std::size_t soundness_bits() { return additive() ? 192 : 180; }
};

// ---- device memory -----------------------------------------------------------------------------------------------------------------
namespace detail {
struct pooled_block {
    void *p = nullptr;
    explicit pooled_block(std::size_t bytes) { check(iopx_pool_alloc(&p, bytes)); }
    pooled_block(const pooled_block &) = delete;
    pooled_block &operator=(const pooled_block &) = delete;
    ~pooled_block() { if (p) iopx_pool_free(p); }
};
} // namespace detail

// `count` elements of elem_bytes in HBM; copies share the block, slice() is a view
template<typename T>
class device_array {
    std::shared_ptr<detail::pooled_block> block_;
    std::size_t offset_ = 0, size_ = 0;
public:
    device_array() {}
    explicit device_array(std::size_t count) : block_(std::make_shared<detail::pooled_block>((count ? count : 1) * sizeof(T))), size_(count) {}
    std::size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    T *data() const { return block_ ? reinterpret_cast<T *>(block_->p) + offset_ : nullptr; }
    device_array slice(std::size_t begin, std::size_t count) const
    {
        if (begin + count > size_) throw std::invalid_argument("device_array::slice out of range");
        device_array r;
        r.block_ = block_; r.offset_ = offset_ + begin; r.size_ = count;
        return r;
    }
    static device_array from_host(const T *src, std::size_t count)
    {
        device_array r(count);
        if (count) check(iopx_memcpy_h2d(r.data(), src, count * sizeof(T)));
        return r;
    }
    static device_array from_host(const std::vector<T> &v) { return from_host(v.data(), v.size()); }
    std::vector<T> to_host(std::size_t count = (std::size_t)-1) const
    {
        if (count == (std::size_t)-1) count = size_;
        if (count > size_) throw std::invalid_argument("device_array::to_host out of range");
        std::vector<T> out(count);
        if (count) check(iopx_memcpy_d2h(out.data(), data(), count * sizeof(T)));
        return out;
    }
    void fill_zero() const { if (size_) check(iopx_memset_dev(data(), 0, size_ * sizeof(T))); }
    void copy_from(const device_array &src) const
    {
        if (src.size() != size_) throw std::invalid_argument("device_array::copy_from: size mismatch");
        if (size_) check(iopx_memcpy_d2d(data(), src.data(), size_ * sizeof(T)));
    }
};

template<typename FieldT>
class device_vector : public device_array<FieldT> {
public:
    device_vector() {}
    explicit device_vector(std::size_t count) : device_array<FieldT>(count) {}
    device_vector(const device_array<FieldT> &a) : device_array<FieldT>(a) {}
    uint64_t *words() const { return reinterpret_cast<uint64_t *>(this->data()); }
    device_vector slice(std::size_t begin, std::size_t count) const { return device_vector(device_array<FieldT>::slice(begin, count)); }
};

} // namespace libiop_amd
