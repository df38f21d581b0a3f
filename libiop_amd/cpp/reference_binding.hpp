// The forwarding bodies of the drop-in boundary, written ONCE, generic over the domain classes.
//
// Every function here is the body of one reference seam expressed through nothing but the accessors the
// reference's own domain classes declare — and with exactly the value categories the reference declares them with:
//
//   affine_subspace<FieldT>::dimension()     std::size_t                     libiop/algebra/field_subset/subspace.hpp:27
//   affine_subspace<FieldT>::basis()         const std::vector<FieldT>&      subspace.hpp:30
//   affine_subspace<FieldT>::shift()         const FieldT   (BY VALUE)       subspace.hpp:61
//   affine_subspace<FieldT>::num_elements()  std::size_t                     subspace.hpp:28
//   multiplicative_coset<FieldT>::generator() FieldT        (BY VALUE)       subgroup.hpp:43
//   multiplicative_coset<FieldT>::shift()    FieldT         (BY VALUE)       subgroup.hpp:102
//   field_subset<FieldT>::shift()            const FieldT   (BY VALUE)       field_subset.hpp:67
//   field_subset<FieldT>::generator()        FieldT         (BY VALUE)       field_subset.hpp:65
//   field_subset<FieldT>::basis()            const std::vector<FieldT>&      field_subset.hpp:68
//
// so the same text compiles against libiop's classes (the stubs of INTEGRATION.md are one-line calls into this header) and
// against the mirror classes of libiop_amd.hpp, which call it too: the bodies a libiop maintainer would bind are the bodies
// tests/cpp/test_shim.cpp runs.  An accessor result is never bound by address: it is copied into a local first.
// tests/test_reference_signatures.py compares the declarations above with the reference's header text (container only).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/libiop_amd.h"

namespace libiop_amd {

// exception types of fft.tcc:333,368 / fri_aux.tcc:33 / merkle_tree.tcc:27-31,98-108 / blake2b.tcc:153-156
inline void check(int rc)
{
    if (rc == IOPX_OK) return;
    const std::string msg = iopx_last_error();
    if (rc == IOPX_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (rc == IOPX_ERR_LOGIC) throw std::logic_error(msg);
    throw std::runtime_error(msg);
}

namespace detail {
template<typename FieldT>
inline const uint64_t *words(const FieldT *p) { return reinterpret_cast<const uint64_t *>(p); }
template<typename FieldT>
inline uint64_t *words(FieldT *p) { return reinterpret_cast<uint64_t *>(p); }
} // namespace detail

namespace binding {

// additive_FFT (fft.hpp:28-32, fft.tcc:39-124); Subspace = affine_subspace<FieldT>
template<typename FieldT, typename Subspace>
std::vector<FieldT> additive_FFT(const std::vector<FieldT> &poly_coeffs, const Subspace &domain)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates fields with libff::gf192's layout (three 64-bit words)");
    const FieldT shift = domain.shift();
    std::vector<FieldT> out(domain.num_elements(), FieldT(0));
    check(iopx_add_fft_gf192(detail::words(poly_coeffs.data()), poly_coeffs.size(), detail::words(domain.basis().data()),
                             domain.dimension(), detail::words(&shift), detail::words(out.data())));
    return out;
}

// additive_IFFT (fft.hpp:34-38, fft.tcc:126-204; size precondition :132)
template<typename FieldT, typename Subspace>
std::vector<FieldT> additive_IFFT(const std::vector<FieldT> &evals, const Subspace &domain)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates fields with libff::gf192's layout (three 64-bit words)");
    if (evals.size() != domain.num_elements()) throw std::invalid_argument("additive_IFFT: evaluation count != domain size");
    const FieldT shift = domain.shift();
    std::vector<FieldT> out(domain.num_elements(), FieldT(0));
    check(iopx_add_ifft_gf192(detail::words(evals.data()), detail::words(domain.basis().data()), domain.dimension(),
                              detail::words(&shift), detail::words(out.data())));
    return out;
}

// multiplicative_FFT (fft.hpp:40-45, fft.tcc:236-317,336-341); Coset = multiplicative_coset<FieldT>
template<typename FieldT, typename Coset>
std::vector<FieldT> multiplicative_FFT(const std::vector<FieldT> &poly_coeffs, const Coset &domain)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates prime fields with libff::edwards_Fr's layout (three Montgomery words)");
    const FieldT g = domain.generator(), shift = domain.shift();
    std::vector<FieldT> out(domain.num_elements());
    check(iopx_mul_fft_fp3(detail::words(poly_coeffs.data()), poly_coeffs.size(), domain.dimension(), detail::words(&g),
                           detail::words(&shift), detail::words(out.data())));
    return out;
}

// multiplicative_IFFT (fft.hpp:47-52, fft.tcc:343-376)
template<typename FieldT, typename Coset>
std::vector<FieldT> multiplicative_IFFT(const std::vector<FieldT> &evals, const Coset &domain)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates prime fields with libff::edwards_Fr's layout (three Montgomery words)");
    if (evals.size() != domain.num_elements()) throw std::invalid_argument("multiplicative_IFFT: evaluation count != domain size");
    const FieldT g = domain.generator(), shift = domain.shift();
    std::vector<FieldT> out(domain.num_elements());
    check(iopx_mul_ifft_fp3(detail::words(evals.data()), domain.dimension(), detail::words(&g), detail::words(&shift),
                            detail::words(out.data())));
    return out;
}

// additive_evaluate_next_f_i_over_entire_domain (fri_aux.tcc:36-103); Domain = affine_subspace<FieldT> or, as the reference
// passes it (fri_aux.hpp:30-35), field_subset<FieldT> — both declare basis() / dimension() / shift() / num_elements()
template<typename FieldT, typename Domain>
std::shared_ptr<std::vector<FieldT>> additive_evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const Domain &f_i_domain, const std::size_t coset_size, const FieldT x_i)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates fields with libff::gf192's layout (three 64-bit words)");
    if (f_i_evals->size() != f_i_domain.num_elements()) throw std::invalid_argument("f_i size != domain size");
    const FieldT shift = f_i_domain.shift();
    auto next = std::make_shared<std::vector<FieldT>>(f_i_domain.num_elements() / coset_size, FieldT(0));
    check(iopx_fri_fold_add_gf192(detail::words(f_i_evals->data()), detail::words(f_i_domain.basis().data()), f_i_domain.dimension(),
                                  detail::words(&shift), coset_size, detail::words(&x_i), detail::words(next->data())));
    return next;
}

// multiplicative_evaluate_next_f_i_over_entire_domain (fri_aux.tcc:106-249); Domain = multiplicative_coset<FieldT> or field_subset<FieldT>
template<typename FieldT, typename Domain>
std::shared_ptr<std::vector<FieldT>> multiplicative_evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const Domain &f_i_domain, const std::size_t coset_size, const FieldT x_i)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates prime fields with libff::edwards_Fr's layout (three Montgomery words)");
    if (f_i_evals->size() != f_i_domain.num_elements()) throw std::invalid_argument("f_i size != domain size");
    const FieldT g = f_i_domain.generator(), shift = f_i_domain.shift();
    auto next = std::make_shared<std::vector<FieldT>>(f_i_domain.num_elements() / coset_size);
    check(iopx_fri_fold_mul_fp3(detail::words(f_i_evals->data()), f_i_domain.dimension(), detail::words(&g), detail::words(&shift),
                                coset_size, detail::words(&x_i), detail::words(next->data())));
    return next;
}

// The BLAKE2b tree of merkle_tree<FieldT, binary_hash_digest>::construct_with_leaves_serialized_by_cosets (merkle_tree.tcc:92-151)
// and compute_inner_nodes (:200-229): fills `nodes` with the (2L-1) 32-byte digests in heap order (inner_nodes_[(L-1)+i] = leaf i).
// `salts` = the zk leaf randomness, L x salt_bytes, or empty for a non-zk tree (blake2b.tcc:126-136).
template<typename FieldT>
void merkle_blake2b_nodes(const std::vector<std::shared_ptr<std::vector<FieldT>>> &leaf_contents, std::size_t coset_serialization_size,
                          bool multiplicative_positions, const std::vector<uint8_t> &salts, std::size_t salt_bytes, std::vector<uint8_t> &nodes)
{
    if (leaf_contents.empty()) throw std::invalid_argument("merkle tree: no leaf contents");
    const std::size_t n = leaf_contents[0]->size(), num_leaves = n / coset_serialization_size;
    std::vector<const void *> cols;
    for (auto &v : leaf_contents) cols.push_back(v->data());
    nodes.assign((2 * num_leaves - 1) * 32, 0);
    check(iopx_merkle_blake2b(cols.data(), cols.size(), sizeof(FieldT), n, coset_serialization_size,
                              multiplicative_positions ? IOPX_DOMAIN_MULTIPLICATIVE : IOPX_DOMAIN_ADDITIVE,
                              salts.empty() ? nullptr : salts.data(), salts.empty() ? 0 : salt_bytes, nodes.data()));
}

// combined_LDT_virtual_oracle<FieldT>::evaluated_contents (ldt_reducer_aux.tcc:39-131) over host vectors; Domain = field_subset<FieldT>.
// `coefficients` = the 2 * num_oracles random coefficients as set_random_coefficients received them (:26-37).
template<typename FieldT, typename Domain>
std::shared_ptr<std::vector<FieldT>> ldt_combine(const std::vector<std::shared_ptr<std::vector<FieldT>>> &constituents,
                                                 const std::vector<std::size_t> &input_oracle_degrees, const std::vector<FieldT> &coefficients,
                                                 const Domain &codeword_domain, bool multiplicative)
{
    static_assert(sizeof(FieldT) == 24, "FieldT must have a 24-byte layout (libff::gf192 / libff::edwards_Fr)");
    const std::size_t n = codeword_domain.num_elements();
    std::vector<void *> bufs(constituents.size() + 1, nullptr);
    auto result = std::make_shared<std::vector<FieldT>>(n);
    const FieldT shift = codeword_domain.shift();
    int rc = IOPX_OK;
    for (std::size_t k = 0; k < bufs.size() && rc == IOPX_OK; ++k) rc = iopx_malloc(&bufs[k], n * sizeof(FieldT));
    for (std::size_t k = 0; k + 1 < bufs.size() && rc == IOPX_OK; ++k) rc = iopx_memcpy_h2d(bufs[k], constituents[k]->data(), n * sizeof(FieldT));
    if (rc == IOPX_OK && multiplicative) {                                                  // ldt_reducer_aux.tcc:104-128
        const FieldT g = codeword_domain.generator();
        rc = iopx_ldt_combine_fp3_dev(bufs.data(), bufs.size() - 1, input_oracle_degrees.data(), detail::words(coefficients.data()),
                                      codeword_domain.dimension(), detail::words(&g), detail::words(&shift), (uint64_t *)bufs.back());
    } else if (rc == IOPX_OK) {                                                             // :78-103
        rc = iopx_ldt_combine_gf192_dev(bufs.data(), bufs.size() - 1, input_oracle_degrees.data(), detail::words(coefficients.data()),
                                        detail::words(codeword_domain.basis().data()), codeword_domain.dimension(), detail::words(&shift),
                                        (uint64_t *)bufs.back());
    }
    if (rc == IOPX_OK) rc = iopx_memcpy_d2h(result->data(), bufs.back(), n * sizeof(FieldT));
    for (void *b : bufs) if (b) iopx_free(b);
    check(rc);
    return result;
}

} // namespace binding
} // namespace libiop_amd
