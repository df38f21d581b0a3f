// Host-side C++ mirror of the reference's template API for the accelerated path.
//
// The reference is header-only C++14 templates over FieldT (no FFI).  This header restates the
// signatures of the seams that the hot path sits behind — same names, argument meaning and exception
// types — and forwards them to the C ABI of include/libiop_amd.h for any FieldT whose in-memory
// layout is libff::gf192's (three little-endian uint64 words; checked with static_assert):
//
//   affine_subspace<FieldT>, multiplicative_coset<FieldT>,
//   field_subset<FieldT> (tagged union, both arms)         libiop/algebra/field_subset/{subspace,subgroup,field_subset}.hpp
//   additive_FFT / additive_IFFT                           libiop/algebra/fft.hpp:28-38   (fft.tcc:39-204)
//   FFT_over_field_subset / IFFT_over_field_subset /
//   IFFT_of_known_degree_over_field_subset                 libiop/algebra/fft.hpp:62-88   (fft.tcc:407-475)
//   evaluate_next_f_i_over_entire_domain                   libiop/protocols/ldt/fri/fri_aux.hpp:23-28
//   merkle_tree<FieldT, hash_digest_type>, leafhash, two_to_one_hash_function,
//   get_leafhash / get_two_to_one_hash, bcs_hash_type      libiop/bcs/merkle_tree.hpp:67-104, bcs/hashing/hashing.hpp:42-53,
//                                                          hash_enum.{hpp,tcc} (BLAKE2b and Poseidon trees)
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
#include <type_traits>
#include <vector>

#include "../../include/libiop_amd.h"
#include "reference_binding.hpp"          // check(), detail::words(), and the forwarding bodies shared with the reference-side stubs

namespace libiop_amd {

template<typename FieldT>
struct is_gf192_layout {
    static const bool value = (sizeof(FieldT) == 24);
};

enum field_subset_type { affine_subspace_type = 1, multiplicative_coset_type = 2 };

// The domain kind of a field: libff::is_additive<FieldT> / libff::is_multiplicative<FieldT> in the reference
// (field_subset.tcc:3-11).  An integration specialises this once per field type, e.g.
//   template<> struct libiop_amd::field_kind<libff::gf192>      { static const field_subset_type type = affine_subspace_type; };
//   template<> struct libiop_amd::field_kind<libff::edwards_Fr> { static const field_subset_type type = multiplicative_coset_type; };
template<typename FieldT> struct field_kind;

namespace detail {
inline std::size_t log2_ceil(std::size_t n) { std::size_t d = 0; while (((std::size_t)1 << d) < n) ++d; return d; }
} // namespace detail

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
    const std::vector<FieldT> &basis() const { return basis_; }                             // subspace.hpp:30
    const FieldT shift() const { return shift_; }                                           // subspace.hpp:61 — by value

    // subspace.tcc:93-108 — default (standard) basis FieldT(1ull << i)
    static affine_subspace shifted_standard_basis(std::size_t dimension, const FieldT &shift)
    {
        std::vector<FieldT> b;
        for (std::size_t i = 0; i < dimension; ++i) b.emplace_back(FieldT((uint64_t)1 << i));
        return affine_subspace(b, shift);
    }
    bool is_standard_basis() const                                                          // subspace.tcc:229-240
    {
        for (std::size_t i = 0; i < basis_.size(); ++i) if (!(basis_[i] == FieldT((uint64_t)1 << i))) return false;
        return true;
    }
    FieldT element_by_index(std::size_t index) const
    {
        if (index >= num_elements()) throw std::invalid_argument("element index out of bounds");
        FieldT r = shift_;
        for (std::size_t i = 0; i < basis_.size(); ++i) if (index & ((std::size_t)1 << i)) r += basis_[i];
        return r;
    }
    FieldT element_outside_of_subset() const                                               // subspace.tcc:219-227
    {
        if (!is_standard_basis()) throw std::invalid_argument("subspace.element_outside_of_subset() is only supported for standard basis");
        return shift_ + FieldT((uint64_t)1 << dimension());
    }
    // subspace.tcc:73-91
    std::size_t coset_index(std::size_t position, std::size_t coset_size) const { return position / coset_size; }
    std::size_t intra_coset_index(std::size_t position, std::size_t coset_size) const { return position % coset_size; }
    std::size_t position_by_coset_indices(std::size_t ci, std::size_t ici, std::size_t coset_size) const { return ci * coset_size + ici; }
};

// libiop/algebra/field_subset/subgroup.hpp — multiplicative_coset: order 2^k, generator g, shift; index i <-> shift * g^i.
// FieldT has libff::edwards_Fr's layout (three uint64 Montgomery words); the subgroup generator multiplicative_generator^
// ((p-1)/order) (subgroup.tcc:55-59) and the few scalar products of the metadata come from the library's host-side helpers, so
// the mirror asks nothing of FieldT beyond its bytes.
template<typename FieldT>
class multiplicative_coset {
    std::size_t order_;
    FieldT g_, shift_;
public:
    multiplicative_coset() : order_(1) {}
    multiplicative_coset(std::size_t order, const FieldT &generator, const FieldT &shift) : order_(order), g_(generator), shift_(shift)
    {
        if (order == 0 || (order & (order - 1))) throw std::invalid_argument("The order of the subgroup must be a power of two.");
    }
    multiplicative_coset(std::size_t order, const FieldT &shift) : order_(order), shift_(shift)        // subgroup.tcc:33-75, 199-215
    {
        static_assert(sizeof(FieldT) == 24, "libiop_amd accelerates prime fields with libff::edwards_Fr's layout");
        if (order == 0 || (order & (order - 1))) throw std::invalid_argument("The order of the subgroup must be a power of two.");
        check(iopx_fp3_subgroup_generator(detail::log2_ceil(order), detail::words(&g_)));
    }
    std::size_t num_elements() const { return order_; }
    std::size_t dimension() const { return detail::log2_ceil(order_); }
    FieldT generator() const { return g_; }                                                 // subgroup.hpp:43 — by value
    FieldT shift() const { return shift_; }                                                 // subgroup.hpp:102 — by value
    FieldT element_by_index(std::size_t index) const                                       // shift * g^index
    {
        if (index >= order_) throw std::invalid_argument("element index out of bounds");
        FieldT p, r;
        check(iopx_fp3_host_pow(detail::words(&g_), index, detail::words(&p)));
        check(iopx_fp3_host_mul(detail::words(&shift_), detail::words(&p), detail::words(&r)));
        return r;
    }
    FieldT element_outside_of_subset() const                                               // subgroup.tcc:311-315
    {
        FieldT gen, r;
        check(iopx_fp3_multiplicative_generator(detail::words(&gen)));
        check(iopx_fp3_host_mul(detail::words(&shift_), detail::words(&gen), detail::words(&r)));
        return r;
    }
    // subgroup.tcc:149-173
    std::size_t reindex_by_subgroup(std::size_t reindex_subgroup_dim, std::size_t index) const
    {
        const std::size_t order_s = (std::size_t)1 << reindex_subgroup_dim, order_g_over_s = (std::size_t)1 << (dimension() - reindex_subgroup_dim);
        if (index < order_s) return index * order_g_over_s;
        const std::size_t i = index - order_s, x = order_g_over_s - 1;
        return i + (i / x) + 1;
    }
    // subgroup.tcc:175-197
    std::size_t coset_index(std::size_t position, std::size_t coset_size) const { return position % (order_ / coset_size); }
    std::size_t intra_coset_index(std::size_t position, std::size_t coset_size) const { return position / (order_ / coset_size); }
    std::size_t position_by_coset_indices(std::size_t ci, std::size_t ici, std::size_t coset_size) const { return ci + ici * (order_ / coset_size); }
};

// libiop/algebra/field_subset/field_subset.hpp — the tagged union over both domain kinds
template<typename FieldT>
class field_subset {
    field_subset_type type_;
    std::shared_ptr<affine_subspace<FieldT>> subspace_;
    std::shared_ptr<multiplicative_coset<FieldT>> coset_;
    bool distributed_ = false;               // multi-GPU provers: vectors over this domain are split over the ranks (dist.hpp); copies keep the mark
    void construct_internal(std::size_t num_elements, const FieldT &shift)                  // field_subset.tcc:33-62
    {
        if (num_elements == 0 || (num_elements & (num_elements - 1))) throw std::invalid_argument("field_subset: size must be a power of two");
        type_ = field_kind<FieldT>::type;
        if (type_ == multiplicative_coset_type) {
            if (shift == FieldT(0)) throw std::invalid_argument("coset_shift was supplied as 0, it was likely intended to be 1");
            coset_ = std::make_shared<multiplicative_coset<FieldT>>(num_elements, shift);
        } else {
            subspace_ = std::make_shared<affine_subspace<FieldT>>(affine_subspace<FieldT>::shifted_standard_basis(detail::log2_ceil(num_elements), shift));
        }
    }
public:
    field_subset() : type_(field_kind<FieldT>::type) {}
    field_subset(const affine_subspace<FieldT> &s) : type_(affine_subspace_type), subspace_(std::make_shared<affine_subspace<FieldT>>(s)) {}
    field_subset(const multiplicative_coset<FieldT> &c) : type_(multiplicative_coset_type), coset_(std::make_shared<multiplicative_coset<FieldT>>(c)) {}
    // field_subset.tcc:3-18: the default domain of that size — shift 0 for binary fields, 1 for prime fields
    explicit field_subset(std::size_t num_elements)
    {
        construct_internal(num_elements, field_kind<FieldT>::type == multiplicative_coset_type ? FieldT(1) : FieldT(0));
    }
    field_subset(std::size_t num_elements, const FieldT &coset_shift) { construct_internal(num_elements, coset_shift); }

    field_subset_type type() const { return type_; }
    bool distributed() const { return distributed_; }
    void set_distributed(bool on) { distributed_ = on; }
    affine_subspace<FieldT> subspace() const                                                // field_subset.hpp:62 — by value
    {
        if (type_ != affine_subspace_type) throw std::invalid_argument("field_subset is not an affine subspace");
        return *subspace_;
    }
    multiplicative_coset<FieldT> coset() const                                              // field_subset.hpp:63 — by value
    {
        if (type_ != multiplicative_coset_type) throw std::invalid_argument("field_subset is not a multiplicative coset");
        return *coset_;
    }
    std::size_t dimension() const { return type_ == affine_subspace_type ? subspace_->dimension() : coset_->dimension(); }
    std::size_t num_elements() const { return type_ == affine_subspace_type ? subspace_->num_elements() : coset_->num_elements(); }
    const std::vector<FieldT> &basis() const                                                // field_subset.hpp:68
    {
        if (type_ != affine_subspace_type) throw std::invalid_argument("field_subset is not an affine subspace");
        return subspace_->basis();
    }
    FieldT generator() const                                                                // field_subset.hpp:65 — by value
    {
        if (type_ != multiplicative_coset_type) throw std::invalid_argument("field_subset is not a multiplicative coset");
        return coset_->generator();
    }
    const FieldT shift() const { return type_ == affine_subspace_type ? subspace_->shift() : coset_->shift(); }   // field_subset.hpp:67 — by value
    FieldT element_by_index(std::size_t i) const { return type_ == affine_subspace_type ? subspace_->element_by_index(i) : coset_->element_by_index(i); }
    FieldT element_outside_of_subset() const                                               // field_subset.tcc:239-252
    {
        return type_ == affine_subspace_type ? subspace_->element_outside_of_subset() : coset_->element_outside_of_subset();
    }
    // field_subset.tcc:130-142
    std::size_t reindex_by_subset(std::size_t reindex_subset_dim, std::size_t index) const
    {
        return type_ == affine_subspace_type ? index : coset_->reindex_by_subgroup(reindex_subset_dim, index);
    }
    // field_subset.tcc:217-237: first log2(order) basis vectors, same shift / the default subgroup of that order, same shift
    field_subset<FieldT> get_subset_of_order(std::size_t order) const
    {
        if (type_ == multiplicative_coset_type) return field_subset(order, shift());
        const std::size_t d = detail::log2_ceil(order);
        if (d > dimension()) throw std::invalid_argument("subset order exceeds the domain");
        return field_subset(affine_subspace<FieldT>(std::vector<FieldT>(basis().begin(), basis().begin() + d), shift()));
    }
    std::size_t coset_index(std::size_t p, std::size_t cs) const { return type_ == affine_subspace_type ? subspace_->coset_index(p, cs) : coset_->coset_index(p, cs); }
    std::size_t intra_coset_index(std::size_t p, std::size_t cs) const
    {
        return type_ == affine_subspace_type ? subspace_->intra_coset_index(p, cs) : coset_->intra_coset_index(p, cs);
    }
    std::size_t position_by_coset_indices(std::size_t ci, std::size_t ici, std::size_t cs) const
    {
        return type_ == affine_subspace_type ? subspace_->position_by_coset_indices(ci, ici, cs) : coset_->position_by_coset_indices(ci, ici, cs);
    }
    std::vector<std::size_t> all_positions_in_coset_i(std::size_t ci, std::size_t cs) const      // field_subset.tcc:187-198
    {
        std::vector<std::size_t> out;
        for (std::size_t i = 0; i < cs; ++i) out.emplace_back(position_by_coset_indices(ci, i, cs));
        return out;
    }
};

// ---- FFT / IFFT (libiop/algebra/fft.hpp) ----------------------------------------------------------
// The bodies are reference_binding.hpp's — the same text the reference-side stubs of INTEGRATION.md call.
template<typename FieldT>
std::vector<FieldT> additive_FFT(const std::vector<FieldT> &poly_coeffs, const affine_subspace<FieldT> &domain)
{
    return binding::additive_FFT<FieldT>(poly_coeffs, domain);
}

template<typename FieldT>
std::vector<FieldT> additive_IFFT(const std::vector<FieldT> &evals, const affine_subspace<FieldT> &domain)
{
    return binding::additive_IFFT<FieldT>(evals, domain);
}

// multiplicative_FFT (fft.tcc:236-317, 336-341)
template<typename FieldT>
std::vector<FieldT> multiplicative_FFT(const std::vector<FieldT> &poly_coeffs, const multiplicative_coset<FieldT> &domain)
{
    return binding::multiplicative_FFT<FieldT>(poly_coeffs, domain);
}

// multiplicative_IFFT (fft.tcc:343-376; size-1 early return :397-401)
template<typename FieldT>
std::vector<FieldT> multiplicative_IFFT(const std::vector<FieldT> &evals, const multiplicative_coset<FieldT> &domain)
{
    return binding::multiplicative_IFFT<FieldT>(evals, domain);
}

// The type-dispatching entry points every protocol calls (fft.tcc:407-433); by-value signatures kept
template<typename FieldT>
std::vector<FieldT> FFT_over_field_subset(const std::vector<FieldT> coeffs, field_subset<FieldT> domain)
{
    if (domain.type() == multiplicative_coset_type) return multiplicative_FFT<FieldT>(coeffs, domain.coset());
    return additive_FFT<FieldT>(coeffs, domain.subspace());
}

template<typename FieldT>
std::vector<FieldT> IFFT_over_field_subset(const std::vector<FieldT> evals, field_subset<FieldT> domain)
{
    if (domain.type() == multiplicative_coset_type) return multiplicative_IFFT<FieldT>(evals, domain.coset());
    return additive_IFFT<FieldT>(evals, domain.subspace());
}

// fft.tcc:435-475: the multiplicative arm interpolates every (n / 2^ceil(log2 degree))-th evaluation over the sub-coset of
// that order (same shift); the additive arm the first 2^ceil(log2 degree) evaluations over the first basis vectors
template<typename FieldT>
std::vector<FieldT> IFFT_of_known_degree_over_field_subset(const std::vector<FieldT> evals, std::size_t degree, field_subset<FieldT> domain)
{
    const std::size_t pow2 = (std::size_t)1 << detail::log2_ceil(degree);
    const field_subset<FieldT> minimal = domain.get_subset_of_order(pow2);
    if (domain.type() == multiplicative_coset_type) {
        std::vector<FieldT> sub;
        const std::size_t freq = domain.num_elements() / pow2;
        for (std::size_t i = 0; i < domain.num_elements(); i += freq) sub.emplace_back(evals[i]);
        return multiplicative_IFFT<FieldT>(sub, minimal.coset());
    }
    const std::vector<FieldT> head(evals.begin(), evals.begin() + pow2);
    return additive_IFFT<FieldT>(head, minimal.subspace());
}

// ---- FRI fold (libiop/protocols/ldt/fri/fri_aux.hpp:23-28; dispatch fri_aux.tcc:5-34) -------------------
// Both arms take the field_subset itself, as fri_aux.tcc:36-41 and :106-111 do.
template<typename FieldT>
std::shared_ptr<std::vector<FieldT>> additive_evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const field_subset<FieldT> &f_i_domain,
    const std::size_t coset_size, const FieldT x_i)
{
    return binding::additive_evaluate_next_f_i_over_entire_domain<FieldT>(f_i_evals, f_i_domain, coset_size, x_i);
}

template<typename FieldT>
std::shared_ptr<std::vector<FieldT>> multiplicative_evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const field_subset<FieldT> &f_i_domain,
    const std::size_t coset_size, const FieldT x_i)
{
    return binding::multiplicative_evaluate_next_f_i_over_entire_domain<FieldT>(f_i_evals, f_i_domain, coset_size, x_i);
}

template<typename FieldT>
std::shared_ptr<std::vector<FieldT>> evaluate_next_f_i_over_entire_domain(
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const field_subset<FieldT> &f_i_domain,
    const std::size_t coset_size, const FieldT x_i)
{
    if (f_i_domain.type() == affine_subspace_type) return additive_evaluate_next_f_i_over_entire_domain<FieldT>(f_i_evals, f_i_domain, coset_size, x_i);
    if (f_i_domain.type() == multiplicative_coset_type) return multiplicative_evaluate_next_f_i_over_entire_domain<FieldT>(f_i_evals, f_i_domain, coset_size, x_i);
    throw std::invalid_argument("f_i_domain is of unsupported domain type");               // fri_aux.tcc:33
}

// ---- hashing abstractions (libiop/bcs/hashing/hashing.hpp:21-53, hash_enum.hpp) -------------------------------------
// The hashes themselves run on the device; the objects the factories return carry WHICH hash a tree uses, so the tree
// constructor takes them exactly where the reference's takes its leafhash / two_to_one_hash_function.
typedef std::string binary_hash_digest;       // hashing.hpp:21
typedef std::string zk_salt_type;

enum bcs_hash_type { blake2b_type = 1, starkware_poseidon_type = 2, high_alpha_poseidon_type = 3 };     // hash_enum.hpp:21-26

template<typename FieldT, typename leaf_hash_type>
class leafhash {
public:
    bcs_hash_type hash_enum;
    std::size_t security_parameter;
    leafhash(bcs_hash_type h, std::size_t sec) : hash_enum(h), security_parameter(sec) {}
};

template<typename hash_type>
struct two_to_one_hash_function {
    bcs_hash_type hash_enum;
    std::size_t security_parameter;
};

namespace detail {
template<typename FieldT, typename hash_type>
inline void check_hash_family(bcs_hash_type hash_enum, std::size_t security_parameter, const char *what)
{
    const bool algebraic = std::is_same<hash_type, FieldT>::value;
    if (algebraic) {
        if (hash_enum != starkware_poseidon_type && hash_enum != high_alpha_poseidon_type)
            throw std::invalid_argument(std::string("bcs_hash_type unknown (algebraic ") + what + ")");      // hash_enum.tcc:71,108,161
        if (security_parameter != 128) throw std::invalid_argument("Poseidon only supported for 128 bit soundness.");
    } else if (hash_enum != blake2b_type) {
        throw std::invalid_argument("bcs_hash_type unknown");
    }
}
} // namespace detail

// hash_enum.tcc:73-127
template<typename FieldT, typename leaf_hash_type>
std::shared_ptr<leafhash<FieldT, leaf_hash_type>> get_leafhash(const bcs_hash_type hash_enum, const std::size_t security_parameter, const std::size_t leaf_size)
{
    (void)leaf_size;
    detail::check_hash_family<FieldT, leaf_hash_type>(hash_enum, security_parameter, "leaf hash");
    return std::make_shared<leafhash<FieldT, leaf_hash_type>>(hash_enum, security_parameter);
}

// hash_enum.tcc:129-171
template<typename hash_type, typename FieldT>
two_to_one_hash_function<hash_type> get_two_to_one_hash(const bcs_hash_type hash_enum, const std::size_t security_parameter)
{
    detail::check_hash_family<FieldT, hash_type>(hash_enum, security_parameter, "two to one hash");
    return two_to_one_hash_function<hash_type>{ hash_enum, security_parameter };
}

// ---- Merkle tree (libiop/bcs/merkle_tree.hpp) -----------------------------------------------------
// libiop/bcs/merkle_tree.hpp:18-36
template<typename hash_digest_type>
struct merkle_tree_set_membership_proof {
    std::vector<hash_digest_type> auxiliary_hashes;
    std::vector<zk_salt_type> randomness_hashes;    // zk salts of the queried leaves, in sorted position order
};

// merkle_tree<FieldT, hash_digest_type> (merkle_tree.hpp:38-140): hash_digest_type = binary_hash_digest selects the BLAKE2b
// kernels, hash_digest_type = FieldT (alt_bn128 Fr layout, 32 bytes) the Poseidon kernels with the parameter set of the
// injected hashers' bcs_hash_type.  Digests are 32 bytes either way.
template<typename FieldT, typename hash_digest_type = binary_hash_digest>
class merkle_tree {
    static const bool algebraic = std::is_same<hash_digest_type, FieldT>::value;
    std::size_t num_leaves_;
    std::shared_ptr<leafhash<FieldT, hash_digest_type>> leaf_hasher_;
    two_to_one_hash_function<hash_digest_type> node_hasher_;
    std::size_t digest_len_bytes_;
    bool make_zk_;
    bool constructed_;
    std::vector<uint8_t> nodes_;               // (2L-1) * 32, heap order (merkle_tree.tcc:114,145)
    std::vector<uint8_t> zk_salts_;
    std::size_t salt_bytes_;

    static hash_digest_type to_digest(const uint8_t *p, std::true_type) { hash_digest_type d; std::memcpy((void *)&d, p, sizeof(d)); return d; }
    static hash_digest_type to_digest(const uint8_t *p, std::false_type) { return hash_digest_type(reinterpret_cast<const char *>(p), 32); }
    static hash_digest_type digest_at(const uint8_t *p) { return to_digest(p, std::integral_constant<bool, algebraic>()); }
public:
    // merkle_tree.tcc:12-33
    merkle_tree(std::size_t num_leaves, const std::shared_ptr<leafhash<FieldT, hash_digest_type>> &leaf_hasher,
                const two_to_one_hash_function<hash_digest_type> &node_hasher, std::size_t digest_len_bytes, bool make_zk,
                std::size_t security_parameter)
        : num_leaves_(num_leaves), leaf_hasher_(leaf_hasher), node_hasher_(node_hasher), digest_len_bytes_(digest_len_bytes), make_zk_(make_zk),
          constructed_(false), salt_bytes_(algebraic ? 32 : (security_parameter * 2 + 7) / 8)
    {
        static_assert(!algebraic || sizeof(FieldT) == 32, "algebraic digests: FieldT must have libff::alt_bn128_Fr's 32-byte layout");
        if (num_leaves < 2 || (num_leaves & (num_leaves - 1)))
            throw std::invalid_argument("Merkle tree size must be a power of two, and at least 2.");
        if (!leaf_hasher) throw std::invalid_argument("merkle_tree: null leaf hasher");
        if (leaf_hasher->hash_enum != node_hasher.hash_enum) throw std::invalid_argument("merkle_tree: leaf and node hashers of different hash types");
        if (!algebraic && digest_len_bytes != 32) throw std::invalid_argument("libiop_amd: only 32-byte BLAKE2b digests are supported");
    }
    // BLAKE2b tree with the factories' defaults: what default_bcs_params wires for binary and 181-bit fields
    explicit merkle_tree(std::size_t num_leaves, std::size_t digest_len_bytes = 32, bool make_zk = false, std::size_t security_parameter = 128)
        : merkle_tree(num_leaves, get_leafhash<FieldT, hash_digest_type>(blake2b_type, security_parameter, 2),
                      get_two_to_one_hash<hash_digest_type, FieldT>(blake2b_type, security_parameter), digest_len_bytes, make_zk, security_parameter) {}

    // zk salts are sampled by the caller (merkle_tree.tcc:36-72 uses libsodium randombytes)
    void set_leaf_randomness(const std::vector<uint8_t> &salts) { zk_salts_ = salts; }

    // merkle_tree.tcc:92-151.  The position map is that of field_subset<FieldT>(num_leaves * coset size) (:118): the field's
    // default domain kind, or the one given explicitly.
    void construct_with_leaves_serialized_by_cosets(const std::vector<std::shared_ptr<std::vector<FieldT>>> &leaf_contents,
                                                    std::size_t coset_serialization_size, int domain_type = -1)
    {
        if (constructed_) throw std::logic_error("Attempting to double-construct a Merkle tree.");
        for (auto &v : leaf_contents)
            if ((v->size() / coset_serialization_size) != num_leaves_)
                throw std::logic_error("Attempting to construct a Merkle tree with a constituent vector of wrong size");
        if (make_zk_ && zk_salts_.size() != num_leaves_ * salt_bytes_) throw std::logic_error("zk Merkle tree without leaf randomness");
        if (domain_type < 0) domain_type = field_kind<FieldT>::type == multiplicative_coset_type ? IOPX_DOMAIN_MULTIPLICATIVE : IOPX_DOMAIN_ADDITIVE;
        std::vector<const void *> ptrs;
        for (auto &v : leaf_contents) ptrs.push_back(v->data());
        nodes_.assign((2 * num_leaves_ - 1) * 32, 0);
        if (algebraic) {
            iopx_poseidon_params pp;
            check(iopx_poseidon_shipped_params((int)leaf_hasher_->hash_enum, 0, &pp));     // get_poseidon_parameters, hash_enum.tcc:12-24
            check(iopx_merkle_poseidon_bn128(&pp, ptrs.data(), ptrs.size(), leaf_contents[0]->size(), coset_serialization_size, domain_type,
                                             make_zk_ ? zk_salts_.data() : nullptr, reinterpret_cast<uint64_t *>(nodes_.data())));
        } else {
            binding::merkle_blake2b_nodes<FieldT>(leaf_contents, coset_serialization_size, domain_type == IOPX_DOMAIN_MULTIPLICATIVE,
                                                  make_zk_ ? zk_salts_ : std::vector<uint8_t>(), salt_bytes_, nodes_);
        }
        constructed_ = true;
    }
    void construct(const std::vector<std::shared_ptr<std::vector<FieldT>>> &leaf_contents)
    {
        construct_with_leaves_serialized_by_cosets(leaf_contents, 1);
    }

    hash_digest_type get_root() const
    {
        if (!constructed_) throw std::logic_error("Attempting to obtain a Merkle tree root without constructing the tree first.");
        return digest_at(nodes_.data());
    }
    hash_digest_type node(std::size_t heap_index) const { return digest_at(nodes_.data() + 32 * heap_index); }
    std::size_t num_leaves() const { return num_leaves_; }

    // merkle_tree.tcc:242-336.  The node array is staged on the device for the call; a prover that keeps its trees in HBM
    // calls iopx_merkle_membership_proof_dev on them directly and only these digests cross PCIe.
    merkle_tree_set_membership_proof<hash_digest_type> get_set_membership_proof(const std::vector<std::size_t> &positions) const
    {
        if (!constructed_) throw std::logic_error("Attempting to obtain a Merkle tree authentication path without constructing the tree first.");
        merkle_tree_set_membership_proof<hash_digest_type> result;
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
        for (std::size_t i = 0; i < count; ++i) result.auxiliary_hashes.emplace_back(digest_at(aux.data() + 32 * i));
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
        return binding::ldt_combine<FieldT>(constituent_oracle_evaluations, input_oracle_degrees_, random_coefficients_, codeword_domain_,
                                            codeword_domain_.type() == multiplicative_coset_type);
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
