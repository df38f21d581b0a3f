// Host-side C++ mirror of the reference's template API for the accelerated path.
//
// The reference is header-only C++14 templates over FieldT (no FFI).  This header restates the
// signatures of the seams that the hot path sits behind — same names, argument meaning and exception
// types — and forwards them to the C ABI of include/libiop_amd.h for any FieldT whose in-memory
// layout is libff::gf192's (three little-endian uint64 words; checked with static_assert):
//
//   affine_subspace<FieldT>, field_subset<FieldT>          libiop/algebra/field_subset/{subspace,field_subset}.hpp
//   additive_FFT / additive_IFFT                           libiop/algebra/fft.hpp:28-38   (fft.tcc:39-204)
//   FFT_over_field_subset / IFFT_over_field_subset /
//   IFFT_of_known_degree_over_field_subset                 libiop/algebra/fft.hpp:62-88   (fft.tcc:407-475)
//   evaluate_next_f_i_over_entire_domain                   libiop/protocols/ldt/fri/fri_aux.hpp:23-28
//   merkle_tree<FieldT, binary_hash_digest>                libiop/bcs/merkle_tree.hpp:67-104 (construct*, get_root)
//   merkle_tree::get_set_membership_proof                  libiop/bcs/merkle_tree.tcc:242-336 (from the device-resident tree)
//   combined_LDT_virtual_oracle<FieldT>                    libiop/protocols/ldt/ldt_reducer_aux.hpp (evaluated_contents)
//   pow_parameters, pow<FieldT, binary_hash_digest>        libiop/bcs/pow.hpp (solve_pow)
//   multiplicative_coset<FieldT>, multiplicative_FFT / _IFFT,
//   multiplicative_evaluate_next_f_i_over_entire_domain    field_subset/subgroup.hpp, fft.hpp:40-52, fri_aux.tcc:106-249
//                                                          (FieldT with libff::edwards_Fr's layout)
//
// INTEGRATION.md shows how libiop's own headers bind to this instead of their CPU bodies.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/libiop_amd.h"

namespace libiop_amd {

inline void check(int rc)
{
    if (rc == IOPX_OK) return;
    const std::string msg = iopx_last_error();
    if (rc == IOPX_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (rc == IOPX_ERR_LOGIC) throw std::logic_error(msg);
    throw std::runtime_error(msg);
}

template<typename FieldT>
struct is_gf192_layout {
    static const bool value = (sizeof(FieldT) == 24);
};

enum field_subset_type { affine_subspace_type = 1, multiplicative_coset_type = 2 };

// libiop/algebra/field_subset/subspace.hpp — basis + shift; index i <-> shift + sum_{bit k of i} basis[k]
template<typename FieldT>
class affine_subspace {
    std::vector<FieldT> basis_;
    FieldT shift_;
public:
    affine_subspace() : shift_(FieldT(0)) {}
    affine_subspace(const std::vector<FieldT> &basis, const FieldT &shift = FieldT(0)) : basis_(basis), shift_(shift) {}

    std::size_t dimension() const { return basis_.size(); }
    std::size_t num_elements() const { return (std::size_t)1 << basis_.size(); }
    const std::vector<FieldT> &basis() const { return basis_; }
    const FieldT &shift() const { return shift_; }

    // subspace.tcc:93-108 — default (standard) basis FieldT(1ull << i)
    static affine_subspace shifted_standard_basis(std::size_t dimension, const FieldT &shift)
    {
        std::vector<FieldT> b;
        for (std::size_t i = 0; i < dimension; ++i) b.emplace_back(FieldT((uint64_t)1 << i));
        return affine_subspace(b, shift);
    }

    FieldT element_by_index(std::size_t index) const
    {
        if (index >= num_elements()) throw std::invalid_argument("element index out of bounds");
        FieldT r = shift_;
        for (std::size_t i = 0; i < basis_.size(); ++i) if (index & ((std::size_t)1 << i)) r += basis_[i];
        return r;
    }
    // subspace.tcc:73-91
    std::size_t coset_index(std::size_t position, std::size_t coset_size) const { return position / coset_size; }
    std::size_t intra_coset_index(std::size_t position, std::size_t coset_size) const { return position % coset_size; }
    std::size_t position_by_coset_indices(std::size_t ci, std::size_t ici, std::size_t coset_size) const { return ci * coset_size + ici; }
};

// libiop/algebra/field_subset/field_subset.hpp — tagged union; only the additive arm is accelerated so far
template<typename FieldT>
class field_subset {
    field_subset_type type_;
    std::shared_ptr<affine_subspace<FieldT>> subspace_;
public:
    field_subset() : type_(affine_subspace_type) {}
    field_subset(const affine_subspace<FieldT> &s) : type_(affine_subspace_type), subspace_(std::make_shared<affine_subspace<FieldT>>(s)) {}
    // field_subset.tcc:3-22 (additive fields): default domain of that size = standard basis, given shift
    field_subset(std::size_t num_elements, const FieldT &shift = FieldT(0)) : type_(affine_subspace_type)
    {
        std::size_t d = 0;
        while (((std::size_t)1 << d) < num_elements) ++d;
        if (((std::size_t)1 << d) != num_elements) throw std::invalid_argument("field_subset: size must be a power of two");
        subspace_ = std::make_shared<affine_subspace<FieldT>>(affine_subspace<FieldT>::shifted_standard_basis(d, shift));
    }

    field_subset_type type() const { return type_; }
    const affine_subspace<FieldT> &subspace() const { return *subspace_; }
    std::size_t dimension() const { return subspace_->dimension(); }
    std::size_t num_elements() const { return subspace_->num_elements(); }
    const std::vector<FieldT> &basis() const { return subspace_->basis(); }
    const FieldT &shift() const { return subspace_->shift(); }
    FieldT element_by_index(std::size_t i) const { return subspace_->element_by_index(i); }

    // field_subset.tcc:217-237: first log2(order) basis vectors, same shift
    field_subset get_subset_of_order(std::size_t order) const
    {
        std::size_t d = 0;
        while (((std::size_t)1 << d) < order) ++d;
        if (d > dimension()) throw std::invalid_argument("subset order exceeds the domain");
        return field_subset(affine_subspace<FieldT>(std::vector<FieldT>(basis().begin(), basis().begin() + d), shift()));
    }
    std::size_t coset_index(std::size_t p, std::size_t cs) const { return subspace_->coset_index(p, cs); }
    std::size_t intra_coset_index(std::size_t p, std::size_t cs) const { return subspace_->intra_coset_index(p, cs); }
    std::size_t position_by_coset_indices(std::size_t ci, std::size_t ici, std::size_t cs) const { return subspace_->position_by_coset_indices(ci, ici, cs); }
    std::vector<std::size_t> all_positions_in_coset_i(std::size_t ci, std::size_t cs) const      // field_subset.tcc:187-198
    {
        std::vector<std::size_t> out;
        for (std::size_t i = 0; i < cs; ++i) out.emplace_back(position_by_coset_indices(ci, i, cs));
        return out;
    }
};

namespace detail {
template<typename FieldT>
inline const uint64_t *words(const FieldT *p) { return reinterpret_cast<const uint64_t *>(p); }
template<typename FieldT>
inline uint64_t *words(FieldT *p) { return reinterpret_cast<uint64_t *>(p); }
} // namespace detail

// ---- FFT / IFFT (libiop/algebra/fft.hpp) ----------------------------------------------------------
template<typename FieldT>
std::vector<FieldT> additive_FFT(const std::vector<FieldT> &poly_coeffs, const affine_subspace<FieldT> &domain)
{
    static_assert(is_gf192_layout<FieldT>::value, "libiop_amd accelerates fields with libff::gf192's layout");
    std::vector<FieldT> out(domain.num_elements(), FieldT(0));
    check(iopx_add_fft_gf192(detail::words(poly_coeffs.data()), poly_coeffs.size(), detail::words(domain.basis().data()),
                             domain.dimension(), detail::words(&domain.shift()), detail::words(out.data())));
    return out;
}

template<typename FieldT>
std::vector<FieldT> additive_IFFT(const std::vector<FieldT> &evals, const affine_subspace<FieldT> &domain)
{
    static_assert(is_gf192_layout<FieldT>::value, "libiop_amd accelerates fields with libff::gf192's layout");
    if (evals.size() != domain.num_elements()) throw std::invalid_argument("additive_IFFT: evaluation count != domain size");
    std::vector<FieldT> out(domain.num_elements(), FieldT(0));
    check(iopx_add_ifft_gf192(detail::words(evals.data()), detail::words(domain.basis().data()), domain.dimension(),
                              detail::words(&domain.shift()), detail::words(out.data())));
    return out;
}

// fft.tcc:414-419, 428-433 — by-value signatures kept
template<typename FieldT>
std::vector<FieldT> FFT_over_field_subset(const std::vector<FieldT> coeffs, field_subset<FieldT> domain)
{
    return additive_FFT<FieldT>(coeffs, domain.subspace());
}

template<typename FieldT>
std::vector<FieldT> IFFT_over_field_subset(const std::vector<FieldT> evals, field_subset<FieldT> domain)
{
    return additive_IFFT<FieldT>(evals, domain.subspace());
}

// fft.tcc:458-475
template<typename FieldT>
std::vector<FieldT> IFFT_of_known_degree_over_field_subset(const std::vector<FieldT> evals, std::size_t degree, field_subset<FieldT> domain)
{
    std::size_t pow2 = 1;
    while (pow2 < degree) pow2 <<= 1;
    const field_subset<FieldT> minimal = domain.get_subset_of_order(pow2);
    const std::vector<FieldT> head(evals.begin(), evals.begin() + pow2);
    return additive_IFFT<FieldT>(head, minimal.subspace());
}

// ---- multiplicative cosets over the 181-bit prime field ------------------------------------------------
// libiop/algebra/field_subset/subgroup.hpp — multiplicative_coset: order 2^k, generator g, shift; index i <-> shift * g^i.
// FieldT must have libff::edwards_Fr's layout (three uint64 Montgomery words).  The generator is supplied by the caller's
// field type (subgroup.tcc:55-59: multiplicative_generator^((p-1)/order)).
template<typename FieldT>
class multiplicative_coset {
    std::size_t order_;
    FieldT g_, shift_;
public:
    multiplicative_coset(std::size_t order, const FieldT &generator, const FieldT &shift) : order_(order), g_(generator), shift_(shift)
    {
        if (order == 0 || (order & (order - 1))) throw std::invalid_argument("The order of the subgroup must be a power of two.");
    }
    std::size_t num_elements() const { return order_; }
    std::size_t dimension() const { std::size_t d = 0; while (((std::size_t)1 << d) < order_) ++d; return d; }
    const FieldT &generator() const { return g_; }
    const FieldT &shift() const { return shift_; }
    // subgroup.tcc:175-197
    std::size_t coset_index(std::size_t position, std::size_t coset_size) const { return position % (order_ / coset_size); }
    std::size_t intra_coset_index(std::size_t position, std::size_t coset_size) const { return position / (order_ / coset_size); }
    std::size_t position_by_coset_indices(std::size_t ci, std::size_t ici, std::size_t coset_size) const { return ci + ici * (order_ / coset_size); }
};

// multiplicative_FFT (fft.tcc:236-317, 336-341)
template<typename FieldT>
std::vector<FieldT> multiplicative_FFT(const std::vector<FieldT> &poly_coeffs, const multiplicative_coset<FieldT> &domain)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates prime fields with libff::edwards_Fr's layout");
    std::vector<FieldT> out(domain.num_elements());
    check(iopx_mul_fft_fp3(detail::words(poly_coeffs.data()), poly_coeffs.size(), domain.dimension(), detail::words(&domain.generator()),
                           detail::words(&domain.shift()), detail::words(out.data())));
    return out;
}

// multiplicative_IFFT (fft.tcc:343-376; size-1 early return :397-401)
template<typename FieldT>
std::vector<FieldT> multiplicative_IFFT(const std::vector<FieldT> &evals, const multiplicative_coset<FieldT> &domain)
{
    static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates prime fields with libff::edwards_Fr's layout");
    if (evals.size() != domain.num_elements()) throw std::invalid_argument("multiplicative_IFFT: evaluation count != domain size");
    std::vector<FieldT> out(domain.num_elements());
    check(iopx_mul_ifft_fp3(detail::words(evals.data()), domain.dimension(), detail::words(&domain.generator()),
                            detail::words(&domain.shift()), detail::words(out.data())));
    return out;
}

// multiplicative_evaluate_next_f_i_over_entire_domain (fri_aux.tcc:106-249)
template<typename FieldT>
std::shared_ptr<std::vector<FieldT>> multiplicative_evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const multiplicative_coset<FieldT> &f_i_domain,
    const std::size_t coset_size, const FieldT x_i)
{
    if (f_i_evals->size() != f_i_domain.num_elements()) throw std::invalid_argument("f_i size != domain size");
    auto next = std::make_shared<std::vector<FieldT>>(f_i_domain.num_elements() / coset_size);
    check(iopx_fri_fold_mul_fp3(detail::words(f_i_evals->data()), f_i_domain.dimension(), detail::words(&f_i_domain.generator()),
                                detail::words(&f_i_domain.shift()), coset_size, detail::words(&x_i), detail::words(next->data())));
    return next;
}

// ---- FRI fold (libiop/protocols/ldt/fri/fri_aux.hpp:23-28) ----------------------------------------
template<typename FieldT>
std::shared_ptr<std::vector<FieldT>> evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const field_subset<FieldT> &f_i_domain,
    const std::size_t coset_size, const FieldT x_i)
{
    static_assert(is_gf192_layout<FieldT>::value, "libiop_amd accelerates fields with libff::gf192's layout");
    if (f_i_domain.type() != affine_subspace_type) throw std::invalid_argument("f_i_domain is of unsupported domain type");
    if (f_i_evals->size() != f_i_domain.num_elements()) throw std::invalid_argument("f_i size != domain size");
    auto next = std::make_shared<std::vector<FieldT>>(f_i_domain.num_elements() / coset_size, FieldT(0));
    check(iopx_fri_fold_add_gf192(detail::words(f_i_evals->data()), detail::words(f_i_domain.basis().data()), f_i_domain.dimension(),
                                  detail::words(&f_i_domain.shift()), coset_size, detail::words(&x_i), detail::words(next->data())));
    return next;
}

// ---- Merkle tree (libiop/bcs/merkle_tree.hpp) -----------------------------------------------------
typedef std::string binary_hash_digest;       // libiop/bcs/hashing/hashing.hpp:21

// libiop/bcs/merkle_tree.hpp:18-36
struct merkle_tree_set_membership_proof {
    std::vector<binary_hash_digest> auxiliary_hashes;
    std::vector<std::string> randomness_hashes;    // zk salts of the queried leaves, in sorted position order
};

template<typename FieldT>
class merkle_tree {
    std::size_t num_leaves_;
    std::size_t digest_len_bytes_;
    bool make_zk_;
    bool constructed_;
    std::vector<uint8_t> nodes_;               // (2L-1) * 32, heap order (merkle_tree.tcc:114,145)
    std::vector<uint8_t> zk_salts_;
    std::size_t salt_bytes_;
public:
    // merkle_tree.tcc:12-33 (leaf / node hashers are BLAKE2b: hash_enum.tcc:112-165 with blake2b_type)
    merkle_tree(std::size_t num_leaves, std::size_t digest_len_bytes = 32, bool make_zk = false, std::size_t security_parameter = 128)
        : num_leaves_(num_leaves), digest_len_bytes_(digest_len_bytes), make_zk_(make_zk), constructed_(false),
          salt_bytes_((security_parameter * 2 + 7) / 8)
    {
        if (num_leaves < 2 || (num_leaves & (num_leaves - 1)))
            throw std::invalid_argument("Merkle tree size must be a power of two, and at least 2.");
        if (digest_len_bytes != 32) throw std::invalid_argument("libiop_amd: only 32-byte BLAKE2b digests are supported");
    }

    // zk salts are sampled by the caller (merkle_tree.tcc:36-72 uses libsodium randombytes)
    void set_leaf_randomness(const std::vector<uint8_t> &salts) { zk_salts_ = salts; }

    // domain_type: position map of the default domain of the leaf_contents' size (field_subset<FieldT>(size), merkle_tree.tcc:118):
    // IOPX_DOMAIN_ADDITIVE for binary fields, IOPX_DOMAIN_MULTIPLICATIVE for prime fields
    void construct_with_leaves_serialized_by_cosets(const std::vector<std::shared_ptr<std::vector<FieldT>>> &leaf_contents,
                                                    std::size_t coset_serialization_size, int domain_type = IOPX_DOMAIN_ADDITIVE)
    {
        if (constructed_) throw std::logic_error("Attempting to double-construct a Merkle tree.");
        for (auto &v : leaf_contents)
            if ((v->size() / coset_serialization_size) != num_leaves_)
                throw std::logic_error("Attempting to construct a Merkle tree with a constituent vector of wrong size");
        if (make_zk_ && zk_salts_.size() != num_leaves_ * salt_bytes_) throw std::logic_error("zk Merkle tree without leaf randomness");
        std::vector<const void *> ptrs;
        for (auto &v : leaf_contents) ptrs.push_back(v->data());
        nodes_.assign((2 * num_leaves_ - 1) * 32, 0);
        check(iopx_merkle_blake2b(ptrs.data(), ptrs.size(), sizeof(FieldT), leaf_contents[0]->size(), coset_serialization_size,
                                  domain_type, make_zk_ ? zk_salts_.data() : nullptr, make_zk_ ? salt_bytes_ : 0, nodes_.data()));
        constructed_ = true;
    }
    void construct(const std::vector<std::shared_ptr<std::vector<FieldT>>> &leaf_contents)
    {
        construct_with_leaves_serialized_by_cosets(leaf_contents, 1);
    }

    binary_hash_digest get_root() const
    {
        if (!constructed_) throw std::logic_error("Attempting to obtain a Merkle tree root without constructing the tree first.");
        return binary_hash_digest(reinterpret_cast<const char *>(nodes_.data()), 32);
    }
    binary_hash_digest node(std::size_t heap_index) const
    {
        return binary_hash_digest(reinterpret_cast<const char *>(nodes_.data()) + 32 * heap_index, 32);
    }
    std::size_t num_leaves() const { return num_leaves_; }

    // merkle_tree.tcc:242-336.  The node array is staged on the device for the call; a prover that keeps its trees in HBM
    // calls iopx_merkle_membership_proof_dev on them directly and only these digests cross PCIe.
    merkle_tree_set_membership_proof get_set_membership_proof(const std::vector<std::size_t> &positions) const
    {
        if (!constructed_) throw std::logic_error("Attempting to obtain a Merkle tree authentication path without constructing the tree first.");
        merkle_tree_set_membership_proof result;
        if (positions.empty()) return result;
        std::vector<std::size_t> S = positions;
        std::sort(S.begin(), S.end());
        S.erase(std::unique(S.begin(), S.end()), S.end());
        void *d_nodes = nullptr;
        check(iopx_malloc(&d_nodes, nodes_.size()));
        std::vector<uint8_t> aux(32 * S.size() * 64);
        std::size_t count = 0;
        int rc = iopx_memcpy_h2d(d_nodes, nodes_.data(), nodes_.size());
        if (rc == IOPX_OK) rc = iopx_merkle_membership_proof_dev((const uint8_t *)d_nodes, num_leaves_, positions.data(), positions.size(), aux.data(), aux.size() / 32, &count);
        iopx_free(d_nodes);
        check(rc);
        for (std::size_t i = 0; i < count; ++i) result.auxiliary_hashes.emplace_back(reinterpret_cast<const char *>(aux.data()) + 32 * i, 32);
        if (make_zk_) for (std::size_t pos : S) result.randomness_hashes.emplace_back(reinterpret_cast<const char *>(zk_salts_.data()) + pos * salt_bytes_, salt_bytes_);
        return result;
    }
};

// ---- LDT reducer (libiop/protocols/ldt/ldt_reducer_aux.hpp) ------------------------------------------------------------
template<typename FieldT>
class combined_LDT_virtual_oracle {
    field_subset<FieldT> codeword_domain_;
    std::vector<std::size_t> input_oracle_degrees_;
    std::vector<FieldT> random_coefficients_;
public:
    combined_LDT_virtual_oracle(const field_subset<FieldT> &codeword_domain, const std::vector<std::size_t> &input_oracle_degrees)
        : codeword_domain_(codeword_domain), input_oracle_degrees_(input_oracle_degrees) {}

    void set_random_coefficients(const std::vector<FieldT> &random_coefficients)        // ldt_reducer_aux.tcc:26-37
    {
        if (random_coefficients.size() != 2 * input_oracle_degrees_.size())
            throw std::invalid_argument("Expected the nunmber of random coefficients to be twice the number of oracles.");
        random_coefficients_ = random_coefficients;
    }

    std::shared_ptr<std::vector<FieldT>> evaluated_contents(                              // ldt_reducer_aux.tcc:39-131
        const std::vector<std::shared_ptr<std::vector<FieldT>>> &constituent_oracle_evaluations) const
    {
        static_assert(is_gf192_layout<FieldT>::value, "FieldT must have libff::gf192's 24-byte layout");
        if (constituent_oracle_evaluations.size() != input_oracle_degrees_.size())
            throw std::invalid_argument("Expected same number of evaluations as in registration.");
        const std::size_t n = constituent_oracle_evaluations[0]->size();
        for (auto &v : constituent_oracle_evaluations) if (v->size() != n) throw std::invalid_argument("Vectors of mismatched size.");
        if (n != codeword_domain_.num_elements()) throw std::invalid_argument("Vectors of mismatched size.");
        std::vector<void *> bufs(constituent_oracle_evaluations.size() + 1, nullptr);
        auto result = std::make_shared<std::vector<FieldT>>(n);
        int rc = IOPX_OK;
        for (std::size_t k = 0; k < bufs.size() && rc == IOPX_OK; ++k) rc = iopx_malloc(&bufs[k], n * sizeof(FieldT));
        for (std::size_t k = 0; k + 1 < bufs.size() && rc == IOPX_OK; ++k)
            rc = iopx_memcpy_h2d(bufs[k], constituent_oracle_evaluations[k]->data(), n * sizeof(FieldT));
        if (rc == IOPX_OK)
            rc = iopx_ldt_combine_gf192_dev(bufs.data(), bufs.size() - 1, input_oracle_degrees_.data(), detail::words(random_coefficients_.data()),
                                            detail::words(codeword_domain_.basis().data()), codeword_domain_.dimension(),
                                            detail::words(&codeword_domain_.shift()), (uint64_t *)bufs.back());
        if (rc == IOPX_OK) rc = iopx_memcpy_d2h(result->data(), bufs.back(), n * sizeof(FieldT));
        for (void *b : bufs) if (b) iopx_free(b);
        check(rc);
        return result;
    }
};

// ---- proof of work (libiop/bcs/pow.hpp) ------------------------------------------------------------------------------
class pow_parameters {
    std::size_t work_parameter_, cost_per_hash_;
public:
    pow_parameters(std::size_t work_parameter = 0, std::size_t cost_per_hash = 1) : work_parameter_(work_parameter), cost_per_hash_(cost_per_hash) {}
    std::size_t pow_bitlen() const                                                        // pow.tcc:21-32
    {
        std::size_t log_cost = 0;
        while (((std::size_t)1 << log_cost) < cost_per_hash_) ++log_cost;
        if (((std::size_t)1 << log_cost) > cost_per_hash_) log_cost -= 1;
        return work_parameter_ - log_cost;
    }
    std::size_t pow_upperbound() const { return 0; }
    std::size_t work_parameter() const { return work_parameter_; }
};

// pow<FieldT, binary_hash_digest> with the BLAKE2b two-to-one hash (pow.tcc:67-103)
class binary_pow {
    pow_parameters parameters_;
public:
    explicit binary_pow(const pow_parameters &params, std::size_t digest_len_bytes = 32) : parameters_(params)
    {
        if (digest_len_bytes != 32) throw std::invalid_argument("libiop_amd: only 32-byte BLAKE2b digests are supported");
    }
    binary_hash_digest solve_pow(const binary_hash_digest &challenge) const
    {
        if (challenge.size() != 32) throw std::invalid_argument("libiop_amd: the proof-of-work challenge is a 32-byte digest");
        uint8_t out[32];
        check(iopx_pow_solve_blake2b(reinterpret_cast<const uint8_t *>(challenge.data()), parameters_.pow_bitlen(), out));
        return binary_hash_digest(reinterpret_cast<const char *>(out), 32);
    }
};

} // namespace libiop_amd
