// Device-resident IOP bookkeeping + BCS transformation, prover side, for C++ callers of the C ABI.
//
// The reference's provers (libiop/snark/*.tcc) drive  iop_protocol<FieldT>  (libiop/iop/iop.{hpp,tcc})  through its subclass
// bcs_prover<FieldT, hash>  (libiop/bcs/bcs_prover.{hpp,tcc}, bcs_common.{hpp,tcc}); oracles are std::vector<FieldT> on the heap
// (libiop/iop/oracles.hpp:22-52).  Here the same registration / round / transcript interface runs with every oracle in HBM:
//
//   device_vector<FieldT>                       the storage behind oracle<FieldT> (pooled device memory, iopx_pool_alloc)
//   oracle<FieldT>, virtual_oracle<FieldT>      iop/oracles.hpp:22-95 (evaluated_contents() downloads lazily)
//   blake2b_hashchain<FieldT>                   bcs/hashing/blake2b.tcc:10-110, 162-257, blake2b.cpp:50-74 (host, 32-byte state)
//   bcs_prover<FieldT>                          iop.tcc:22-433 registration + rounds, bcs_common.tcc:399-696 one tree per
//                                               (round, domain), bcs_prover.tcc:23-98 round end + proof of work, :136-233 transcript
//   bcs_transformation_transcript<FieldT>       bcs/bcs_common.hpp:36-106
//
// Only 32-byte roots, O(log n) challenges and, at the end, the queried values and authentication paths cross PCIe
// (iopx_transfer_stats counts them).  Non-zk, BLAKE2b digests (what default_bcs_params wires for both accelerated fields).
// FieldT is any 24-byte type with libff::gf192's or libff::edwards_Fr's layout; field_kind<FieldT> says which.
#pragma once
#include <algorithm>
#include <functional>
#include <map>
#include <set>

#include "dist.hpp"

namespace libiop_amd {

// ---- oracles (libiop/iop/oracles.hpp) -------------------------------------------------------------------------------------------
template<typename FieldT>
class oracle {
    device_vector<FieldT> device_contents_;
    mutable std::shared_ptr<std::vector<FieldT>> host_contents_;
public:
    oracle() {}
    oracle(const device_vector<FieldT> &contents) : device_contents_(contents) {}
    oracle(const std::vector<FieldT> &evaluated_contents) : device_contents_(device_vector<FieldT>::from_host(evaluated_contents)) {}
    const device_vector<FieldT> &device_contents() const { return device_contents_; }
    // oracles.hpp:41-44; the one place a whole oracle crosses PCIe, and only when a caller asks for host data
    const std::shared_ptr<std::vector<FieldT>> &evaluated_contents() const
    {
        if (!host_contents_) host_contents_ = std::make_shared<std::vector<FieldT>>(device_contents_.to_host());
        return host_contents_;
    }
    std::size_t size() const { return device_contents_.size(); }
};

template<typename FieldT>
class virtual_oracle {
public:
    virtual ~virtual_oracle() {}
    // oracles.hpp:56-95: the prover side needs the whole-domain evaluation only
    virtual device_vector<FieldT> evaluated_contents(const std::vector<device_vector<FieldT>> &constituent_oracle_evaluations) const = 0;
    // The same pointwise map over a sub-domain D of the oracle's domain: D's evaluations of the constituents in, D's evaluations of the
    // oracle out (bcs_prover::get_oracle_evaluations_over_head).  D is never distributed.
    virtual bool restrictable() const { return false; }
    virtual std::size_t smallest_window() const { return 2; }          // the fewest points of a window this oracle can be evaluated over
    virtual device_vector<FieldT> evaluated_contents_over(const field_subset<FieldT> &, const std::vector<device_vector<FieldT>> &) const
    {
        throw std::logic_error("this virtual oracle is only evaluated over its whole domain");
    }
};

// ---- handles (libiop/iop/iop.hpp:27-196) ----------------------------------------------------------------------------------------
struct domain_handle { std::size_t id; };
struct oracle_handle { std::size_t id; bool is_virtual; };
typedef oracle_handle oracle_handle_ptr;
struct prover_message_handle { std::size_t id; };
struct verifier_random_message_handle { std::size_t id; };
struct query_position_handle { std::size_t id; bool random; };
struct query_handle { std::size_t id; };

// ---- Fiat-Shamir hashchain (bcs/hashing/blake2b.tcc) ----------------------------------------------------------------------------
typedef std::string hash_digest;        // 32 raw bytes (binary_hash_digest)

template<typename FieldT>
class blake2b_hashchain {
    uint8_t state_[32];
    uint64_t squeeze_index_ = 0;
public:
    // absorb() never mixes its input in (reference quirk F8), so a challenge is a function of the round structure alone.  bcs_prover relies on that:
    // it queues the roots' read-backs and builds trees beside the next round (defer_roots / merkle_aside).  A hashchain that really absorbs — the
    // reference's algebraic sponge chain — must set this to true, which switches both off (bcs_prover::finish_round).
    static const bool absorbs_input = false;
    blake2b_hashchain() { std::memset(state_, ' ', 32); }                                     // blake2b.tcc:17
    // :50-61 — hashes the first digest_len bytes of state || input: the state advances to BLAKE2b-256(state) whatever is absorbed
    // (reference behaviour F8 of SURVEY.md, reproduced because the verifier does the same)
    void absorb() { uint8_t next[32]; check(iopx_blake2b_host(next, 32, state_, 32, nullptr, 0)); std::memcpy(state_, next, 32); }
    // :76-86, :162-257 — element i of squeeze number q = keyed BLAKE2b(state || q, key = i), written raw into the element; prime
    // field: bits above the modulus MSB cleared, retried with key += num_elements until below p (the bytes ARE mont_repr)
    std::vector<FieldT> squeeze(std::size_t num_elements)
    {
        ++squeeze_index_;
        uint8_t msg[40];
        std::memcpy(msg, state_, 32);
        std::memcpy(msg + 32, &squeeze_index_, 8);
        std::vector<FieldT> out(num_elements);
        uint64_t modulus[3] = { 0, 0, 0 };
        const bool additive = field_host<FieldT>::additive();
        if (!additive) check(iopx_fp3_modulus(modulus));
        for (std::size_t i = 0; i < num_elements; ++i) {
            uint64_t key = i, w[3];
            for (;;) {
                check(iopx_blake2b_host(reinterpret_cast<uint8_t *>(w), 24, msg, 40, &key, 8));
                if (additive) break;
                key += num_elements;
                int top = 63;
                while (top > 0 && !((modulus[2] >> top) & 1)) --top;                         // modulus MSB within the top limb
                w[2] &= (top == 63) ? ~0ull : ((1ull << (top + 1)) - 1);
                bool below = false;
                for (int k = 2; k >= 0; --k) if (w[k] != modulus[k]) { below = w[k] < modulus[k]; break; }
                if (below) break;
            }
            out[i] = field_host<FieldT>::from_words(w);
        }
        return out;
    }
    // :105-110 — one squeezed element hashed to a digest (blake2b_field_element_hash, :140-160)
    hash_digest squeeze_root_type()
    {
        const std::vector<FieldT> x = squeeze(1);
        uint8_t d[32];
        check(iopx_blake2b_host(d, 32, &x[0], 24, nullptr, 0));
        return hash_digest(reinterpret_cast<const char *>(d), 32);
    }
    // :88-105 + blake2b.cpp:50-74
    std::vector<std::size_t> squeeze_query_positions(std::size_t num_positions, std::size_t range_of_positions)
    {
        if (range_of_positions & (range_of_positions - 1)) throw std::invalid_argument("upper_bound must be a power of two.");
        std::vector<std::size_t> out;
        for (std::size_t i = 0; i < num_positions; ++i) {
            ++squeeze_index_;
            uint64_t v = 0;
            check(iopx_blake2b_host(reinterpret_cast<uint8_t *>(&v), 8, state_, 32, &squeeze_index_, 8));
            out.push_back((std::size_t)(v % range_of_positions));
        }
        return out;
    }
};

// ---- a BCS Merkle tree resident in HBM (bcs/merkle_tree.tcc:92-229) -------------------------------------------------------------
// Over a distributed domain (dist.hpp) the tree is N sub-trees, one per rank over a contiguous run of leaves, under log2 N top levels:
// the N sub-roots are all-gathered (32 bytes each) and the top levels are hashed on every rank.  Subspaces: the rank's block of every
// oracle IS a contiguous run of leaves.  Cosets: local leaf l' is global leaf rank + N l' (the sub-coset's own coset structure), so the
// leaf digests are exchanged once (all-to-all, 32 bytes per leaf instead of coset_size x oracles x 24) into contiguous runs.
class device_merkle_tree {
    device_array<uint8_t> nodes_;              // the whole tree, or this rank's sub-tree
    std::size_t num_leaves_ = 0;               // leaves of nodes_
    bool distributed_ = false;
    std::size_t global_leaves_ = 0;
    device_array<uint8_t> top_nodes_;          // heap order over the N sub-roots (2N - 1 digests), replicated
public:
    device_merkle_tree() {}
    template<typename FieldT>
    device_merkle_tree(const std::vector<device_vector<FieldT>> &oracles, const field_subset<FieldT> &domain, std::size_t coset_size)
        : distributed_(domain.distributed()), global_leaves_(domain.num_elements() / coset_size)
    {
        std::vector<const void *> ptrs;
        for (auto &o : oracles) ptrs.push_back(o.data());
        const int type = domain.type() == affine_subspace_type ? IOPX_DOMAIN_ADDITIVE : IOPX_DOMAIN_MULTIPLICATIVE;
        if (!distributed_) {
            num_leaves_ = global_leaves_;
            nodes_ = device_array<uint8_t>((2 * num_leaves_ - 1) * 32);
            check(iopx_merkle_blake2b_dev(ptrs.data(), ptrs.size(), sizeof(FieldT), domain.num_elements(), coset_size, type, nullptr, 0, nodes_.data()));
            return;
        }
        const dist::context &c = dist::ctx();
        const std::size_t W = c.world, n_local = domain.num_elements() / W;
        num_leaves_ = global_leaves_ / W;
        nodes_ = device_array<uint8_t>((2 * num_leaves_ - 1) * 32);
        if (domain.type() == affine_subspace_type) {
            check(iopx_merkle_blake2b_dev(ptrs.data(), ptrs.size(), sizeof(FieldT), n_local, coset_size, type, nullptr, 0, nodes_.data()));
        } else {
            if (num_leaves_ % W) throw std::logic_error("fewer leaves per rank than ranks");
            check(iopx_merkle_leaves_blake2b_dev(ptrs.data(), ptrs.size(), sizeof(FieldT), n_local, coset_size, type, nullptr, 0, nodes_.data()));
            if (W > 1) {
                // chunk q of this rank's leaf digests holds the leaves that fall into rank q's run; entry u of the chunk received from rank s
                // is global leaf rank L/N + s + N u -> position s + N u of the run
                const device_array<uint8_t> mine(num_leaves_ * 32), got(num_leaves_ * 32);
                check(iopx_memcpy_d2d(mine.data(), nodes_.data() + (num_leaves_ - 1) * 32, num_leaves_ * 32));
                check(iopx_comm_all_to_all_dev(c.comm, mine.data(), got.data(), (num_leaves_ / W) * 32));
                check(iopx_interleave_dev(got.data(), W, num_leaves_ / W, 32, nodes_.data() + (num_leaves_ - 1) * 32));
            }
            check(iopx_merkle_inner_blake2b_dev(nodes_.data(), num_leaves_));
        }
        top_nodes_ = device_array<uint8_t>((2 * W - 1) * 32);
        check(iopx_comm_all_gather_dev(c.comm, nodes_.data(), top_nodes_.data() + (W - 1) * 32, 32));
        if (W > 1) check(iopx_merkle_inner_blake2b_dev(top_nodes_.data(), W));
    }
    std::size_t num_leaves() const { return distributed_ ? global_leaves_ : num_leaves_; }
    hash_digest get_root() const                                                              // merkle_tree.tcc:231-240
    {
        uint8_t d[32];
        check(iopx_memcpy_d2h(d, distributed_ ? top_nodes_.data() : nodes_.data(), 32));
        return hash_digest(reinterpret_cast<const char *>(d), 32);
    }
    // the same read-back queued into dst (32 bytes that outlive the window) inside an iopx_defer_downloads window, immediate outside one
    void get_root_into(uint8_t *dst) const { check(iopx_memcpy_d2h_deferrable(dst, distributed_ ? top_nodes_.data() : nodes_.data(), 32)); }
    // the auxiliary hashes of the pruned multi-membership proof as raw bytes (32 each); inside an iopx_defer_downloads window the buffer is
    // filled by iopx_defer_downloads_end
    std::vector<uint8_t> get_set_membership_proof_bytes(const std::vector<std::size_t> &leaf_positions) const
    {
        if (leaf_positions.empty()) return std::vector<uint8_t>();
        if (distributed_) return distributed_membership_proof_bytes(leaf_positions);
        std::size_t depth = 0;
        while (((std::size_t)1 << depth) < num_leaves_) ++depth;
        std::vector<uint8_t> aux(32 * leaf_positions.size() * (depth + 1));
        std::size_t count = 0;
        check(iopx_merkle_membership_proof_dev(nodes_.data(), num_leaves_, leaf_positions.data(), leaf_positions.size(), aux.data(), aux.size() / 32, &count));
        aux.resize(32 * count);                 // shrinking keeps the storage (and the address the library writes to)
        return aux;
    }
    static std::vector<hash_digest> digests_of(const std::vector<uint8_t> &bytes)
    {
        std::vector<hash_digest> out;
        for (std::size_t i = 0; i + 32 <= bytes.size(); i += 32) out.emplace_back(reinterpret_cast<const char *>(bytes.data()) + i, 32);
        return out;
    }
    std::vector<hash_digest> get_set_membership_proof(const std::vector<std::size_t> &leaf_positions) const      // :242-336
    {
        return digests_of(get_set_membership_proof_bytes(leaf_positions));
    }
private:
    // merkle_tree::get_set_membership_proof over the distributed tree: the index walk on the host (the same on every rank); every rank
    // writes the auxiliary nodes it owns into a zeroed (count, 32) device buffer — the nodes of the top levels come from the replicated top
    // table and are written by rank 0 — and one all-reduce (sum: exactly one owner per row) completes it everywhere.
    std::vector<uint8_t> distributed_membership_proof_bytes(const std::vector<std::size_t> &leaf_positions) const
    {
        const dist::context &c = dist::ctx();
        const std::vector<std::size_t> idx = dist::membership_proof_node_indices(global_leaves_, leaf_positions);
        std::vector<uint8_t> aux(32 * idx.size());
        if (idx.empty()) return aux;
        const device_array<uint8_t> buf(aux.size());
        buf.fill_zero();
        gather_owned_proof_nodes(idx, buf.data());
        check(iopx_comm_all_reduce_u64_dev(c.comm, buf.data(), aux.size() / 8, IOPX_COMM_SUM));
        check(iopx_memcpy_d2h_deferrable(aux.data(), buf.data(), aux.size()));
        return aux;
    }
public:
    bool distributed() const { return distributed_; }
    // the two halves of the above for a caller that completes the rows of SEVERAL trees with one all-reduce (bcs_prover::extract_queries): the
    // heap indices of the auxiliary nodes (the same on every rank), and this rank's nodes written into a zeroed (indices, 32) device buffer
    std::vector<std::size_t> distributed_proof_nodes(const std::vector<std::size_t> &leaf_positions) const
    {
        return dist::membership_proof_node_indices(global_leaves_, leaf_positions);
    }
    void gather_owned_proof_nodes(const std::vector<std::size_t> &idx, uint8_t *d_rows) const
    {
        const dist::context &c = dist::ctx();
        std::vector<uint64_t> rows, local_nodes, top_rows, top_nodes;
        for (std::size_t row = 0; row < idx.size(); ++row) {
            const std::size_t node = idx[row];
            std::size_t depth = 0;
            while ((((std::size_t)2 << depth) - 1) <= node) ++depth;                        // node is at depth floor(log2(node + 1))
            if (depth <= c.log_world) {
                if (c.rank == 0) { top_rows.push_back(row); top_nodes.push_back(node); }
                continue;
            }
            const std::size_t j = node - (((std::size_t)1 << depth) - 1), loc_depth = depth - c.log_world;
            if ((j >> loc_depth) == c.rank) {
                rows.push_back(row);
                local_nodes.push_back((((std::size_t)1 << loc_depth) - 1) + (j & (((std::size_t)1 << loc_depth) - 1)));
            }
        }
        const void *src = nodes_.data(), *top = top_nodes_.data();
        if (!rows.empty()) check(iopx_gather_rows_dev(&src, 1, 32, local_nodes.data(), rows.data(), rows.size(), d_rows));
        if (!top_rows.empty()) check(iopx_gather_rows_dev(&top, 1, 32, top_nodes.data(), top_rows.data(), top_rows.size(), d_rows));
    }
};

// ---- bcs_transformation_transcript (bcs/bcs_common.hpp:36-106) -------------------------------------------------------------------
template<typename FieldT>
struct bcs_transformation_transcript {
    std::vector<std::vector<FieldT>> prover_messages_;
    std::vector<hash_digest> MT_roots_;
    std::vector<std::vector<std::size_t>> query_positions_;          // per tree, sorted
    std::vector<std::vector<std::size_t>> MT_leaf_positions_;        // per tree, sorted
    std::vector<std::vector<std::vector<FieldT>>> query_responses_;  // per tree: [position][oracle]
    std::vector<std::vector<hash_digest>> MT_set_membership_proofs_; // per tree: auxiliary hashes
    hash_digest proof_of_work_;

    // The canonical byte form the parity tests compare (the reference's own serialisation is text, bcs_common.tcc:96-390): counts
    // and positions as 8-byte little-endian integers, elements and digests raw, in the field order of this struct.
    std::string serialize() const
    {
        std::string out;
        auto u64 = [&](uint64_t v) { out.append(reinterpret_cast<const char *>(&v), 8); };
        u64(prover_messages_.size());
        for (auto &m : prover_messages_) { u64(m.size()); out.append(reinterpret_cast<const char *>(m.data()), m.size() * sizeof(FieldT)); }
        u64(MT_roots_.size());
        for (auto &r : MT_roots_) out.append(r);
        for (std::size_t t = 0; t < query_positions_.size(); ++t) {
            u64(query_positions_[t].size());
            for (std::size_t p : query_positions_[t]) u64(p);
            u64(MT_leaf_positions_[t].size());
            for (std::size_t p : MT_leaf_positions_[t]) u64(p);
            const auto &resp = query_responses_[t];
            u64(resp.empty() ? 0 : resp[0].size());
            for (auto &row : resp) out.append(reinterpret_cast<const char *>(row.data()), row.size() * sizeof(FieldT));
            u64(MT_set_membership_proofs_[t].size());
            for (auto &d : MT_set_membership_proofs_[t]) out.append(d);
        }
        out.append(proof_of_work_);
        return out;
    }
};

// bcs_prover_index (bcs/bcs_common.hpp): the index round's oracles and their Merkle trees, device resident, plus what the protocol's
// indexer wants the prover to find next to them (iop_prover_index::all_oracle_evals_; here also the index evaluations over K)
template<typename FieldT>
struct bcs_prover_index {
    std::vector<device_vector<FieldT>> oracles;
    std::vector<device_merkle_tree> trees;
    std::vector<hash_digest> roots;
    std::vector<std::vector<FieldT>> prover_messages;
    std::vector<std::vector<device_vector<FieldT>>> index_evals_over_K;
    // windows of the index oracles (bcs_prover::get_oracle_evaluations_over_window) that earlier proofs gathered: part of the index from then on
    mutable std::map<std::size_t, std::vector<std::pair<dist::window, device_vector<FieldT>>>> oracle_windows;
};
struct bcs_verifier_index {                     // bcs_indexer::get_verifier_index (bcs_indexer.tcc:67-77)
    std::vector<hash_digest> index_MT_roots_;
};

// ---- the round driver ----------------------------------------------------------------------------------------------------------
// Without interactions registered after the index round it is the bcs_indexer (bcs/bcs_indexer.tcc): signal_index_submissions_done
// then Merkleises the submitted index oracles.
template<typename FieldT>
class bcs_prover {
public:
    typedef std::function<std::size_t(const std::vector<std::size_t> &)> position_calculator;
private:
    struct oracle_registration { std::size_t domain, degree; std::string name; };
    struct virtual_registration {
        std::size_t domain, degree;
        std::vector<oracle_handle> constituents;
        std::shared_ptr<virtual_oracle<FieldT>> contents;
        bool cache;
    };
    struct tree_info { std::size_t round, domain; std::vector<std::size_t> oracle_ids; };

    std::size_t pow_bitlen_;
    const bcs_prover_index<FieldT> *index_ = nullptr;                                        // bcs_prover.tcc:12-21
    bool is_holographic_ = false;
    blake2b_hashchain<FieldT> hashchain_;
    // registrations (iop.tcc:22-263)
    std::vector<field_subset<FieldT>> domains_;
    std::vector<oracle_registration> oracle_regs_;
    std::vector<virtual_registration> virtual_regs_;
    std::vector<std::size_t> prover_message_sizes_, verifier_message_sizes_;
    std::vector<std::size_t> num_oracles_at_end_of_round_, num_prover_messages_at_end_of_round_, num_verifier_messages_at_end_of_round_;
    std::vector<std::size_t> round_params_;
    bool from_prover_ = false, sealed_ = false;
    std::size_t num_interaction_rounds_ = 0;
    std::vector<std::size_t> random_position_domains_;
    std::vector<std::pair<std::vector<query_position_handle>, position_calculator>> deterministic_positions_;
    std::vector<std::pair<oracle_handle, query_position_handle>> queries_;
    // run state
    std::vector<device_vector<FieldT>> oracles_;
    std::vector<bool> oracle_submitted_, message_submitted_;
    std::vector<std::vector<FieldT>> prover_messages_;
    std::size_t num_prover_rounds_done_ = 0, processed_MTs_ = 0;
    std::vector<device_merkle_tree> MT_trees_;
    std::vector<hash_digest> MT_roots_;
    std::vector<tree_info> MT_info_;
    std::map<std::size_t, std::vector<FieldT>> verifier_random_messages_;
    std::map<std::size_t, device_vector<FieldT>> virtual_contents_cache_;
    std::size_t head_hint_ = 0;
    dist::window_collector<FieldT> collector_, *outer_collector_ = nullptr;
    std::map<std::size_t, std::vector<std::pair<dist::window, device_vector<FieldT>>>> window_cache_, real_window_cache_;
    hash_digest pow_answer_;

    void assert_can_register(std::size_t domain, std::size_t degree) const
    {
        if (sealed_) throw std::logic_error("attempted to register an oracle after interactive registrations sealed");
        if (domain >= domains_.size()) throw std::invalid_argument("domain not registered");
        if (degree >= domains_[domain].num_elements()) throw std::invalid_argument("attempting to register oracle whose degree exceeds domain size");
    }
    void update_rounds_and_direction(bool from_prover)                                       // iop.tcc:36-63
    {
        if (from_prover_ == from_prover) return;
        if (from_prover_) {
            num_oracles_at_end_of_round_.push_back(oracle_regs_.size());
            num_prover_messages_at_end_of_round_.push_back(prover_message_sizes_.size());
            ++num_interaction_rounds_;
        } else {
            num_verifier_messages_at_end_of_round_.push_back(verifier_message_sizes_.size());
        }
        from_prover_ = from_prover;
    }
    // iop.tcc:801-820: domain -> oracle ids of the round, domains in handle order (std::map)
    std::map<std::size_t, std::vector<std::size_t>> oracles_in_round_by_domain(std::size_t round) const
    {
        const std::size_t begin = round == 0 ? 0 : num_oracles_at_end_of_round_[round - 1];
        std::map<std::size_t, std::vector<std::size_t>> mapping;
        for (std::size_t oid = begin; oid < num_oracles_at_end_of_round_[round]; ++oid) mapping[oracle_regs_[oid].domain].push_back(oid);
        return mapping;
    }
    std::size_t get_round_parameters(std::size_t round) const { return round < round_params_.size() ? round_params_[round] : 1; }

    void run_hashchain_for_round(std::size_t round, std::size_t num_roots)                   // bcs_common.tcc:550-614
    {
        for (std::size_t i = 0; i < num_roots; ++i) hashchain_.absorb();
        hashchain_.absorb();                                                                 // the round's prover messages
        const std::size_t start = num_verifier_messages_at_end_of_round_[round];
        const std::size_t end = round == num_interaction_rounds_ - 1 ? 0 : num_verifier_messages_at_end_of_round_[round + 1];
        for (std::size_t i = start; i < end; ++i) verifier_random_messages_[i] = hashchain_.squeeze(verifier_message_sizes_[i]);
    }

    std::size_t obtain_query_position(const query_position_handle &h, std::map<std::size_t, std::size_t> &random_cache, std::map<std::size_t, std::size_t> &det_cache)
    {
        if (h.random) {
            auto it = random_cache.find(h.id);
            if (it != random_cache.end()) return it->second;
            const std::size_t n = domains_[random_position_domains_[h.id]].num_elements();
            return random_cache[h.id] = hashchain_.squeeze_query_positions(1, n)[0];         // bcs_common.tcc:536-548
        }
        auto it = det_cache.find(h.id);
        if (it != det_cache.end()) return it->second;
        std::vector<std::size_t> seeds;
        for (auto &s : deterministic_positions_[h.id].first) seeds.push_back(obtain_query_position(s, random_cache, det_cache));
        return det_cache[h.id] = deterministic_positions_[h.id].second(seeds);
    }
    // get_oracle_evaluation_at_point with record = true (iop.tcc:669-714): a query to a virtual oracle touches its constituents
    // the proper oracles under a virtual one, each once: resolved on first use (the query phase asks for the same few virtual oracles at
    // thousands of positions, on the proof's critical path)
    mutable std::map<std::size_t, std::vector<std::size_t>> flat_constituents_;
    void flatten(const oracle_handle &h, std::set<std::size_t> &out) const
    {
        if (!h.is_virtual) { out.insert(h.id); return; }
        for (auto &c : virtual_regs_[h.id].constituents) flatten(c, out);
    }
    void record(const oracle_handle &h, std::size_t position, std::vector<std::vector<std::size_t>> &positions_by_oracle) const
    {
        if (!h.is_virtual) { positions_by_oracle[h.id].push_back(position); return; }
        auto it = flat_constituents_.find(h.id);
        if (it == flat_constituents_.end()) {
            std::set<std::size_t> ids;
            flatten(h, ids);
            it = flat_constituents_.emplace(h.id, std::vector<std::size_t>(ids.begin(), ids.end())).first;
        }
        for (std::size_t id : it->second) positions_by_oracle[id].push_back(position);
    }

public:
    explicit bcs_prover(std::size_t pow_work_parameter, const bcs_prover_index<FieldT> *index = nullptr) : pow_bitlen_(pow_work_parameter), index_(index)
    {
        outer_collector_ = dist::active_collector<FieldT>();          // this prover's transforms leave their wanted windows with it (dist::window_collector)
        dist::active_collector<FieldT>() = &collector_;
    }
    bcs_prover(const bcs_prover &) = delete;                          // (the user-declared destructor suppressed the moves; copies were never meant)
    bcs_prover &operator=(const bcs_prover &) = delete;
    ~bcs_prover()
    {
        dist::active_collector<FieldT>() = outer_collector_;
        (void)iopx_side_stream_join();                                // trees still being built beside the rounds read this object's oracles (an unwinding proof)
    }

    // ---- registration ----
    domain_handle register_domain(const field_subset<FieldT> &S) { domains_.push_back(S); return domain_handle{ domains_.size() - 1 }; }
    field_subset<FieldT> get_domain(const domain_handle &h) const { return domains_[h.id]; }      // by value: later registrations may move the table
    oracle_handle register_oracle(const std::string &name, const domain_handle &domain, std::size_t degree, bool make_zk)
    {
        assert_can_register(domain.id, degree);
        if (make_zk) throw std::invalid_argument("zero-knowledge oracles are out of scope (salts and masks are not reproducible)");
        update_rounds_and_direction(true);
        if (is_holographic_ && num_interaction_rounds_ == 0) throw std::invalid_argument("Cannot register non-index oracles in round 0 of a holographic IOP");
        oracle_regs_.push_back({ domain.id, degree, name });
        oracles_.emplace_back();
        oracle_submitted_.push_back(false);
        return oracle_handle{ oracle_regs_.size() - 1, false };
    }
    oracle_handle register_index_oracle(const domain_handle &domain, std::size_t degree)     // iop.tcc:106-125
    {
        assert_can_register(domain.id, degree);
        if (num_prover_rounds_done_ != 0) throw std::invalid_argument("index oracles must be created in the 0th round");
        update_rounds_and_direction(true);
        is_holographic_ = true;
        oracle_regs_.push_back({ domain.id, degree, "index" });
        oracles_.emplace_back();
        oracle_submitted_.push_back(false);
        return oracle_handle{ oracle_regs_.size() - 1, false };
    }
    void signal_index_registrations_done()                                                   // iop.tcc:377-386
    {
        if (!is_holographic_ || num_interaction_rounds_ != 0) throw std::invalid_argument("Should only be used to end round 0 of a holographic IOP");
        update_rounds_and_direction(false);
    }
    oracle_handle register_virtual_oracle(const domain_handle &domain, std::size_t degree, const std::vector<oracle_handle> &constituents,
                                          const std::shared_ptr<virtual_oracle<FieldT>> &contents, bool cache_evaluated_contents = false)
    {
        assert_can_register(domain.id, degree);
        virtual_regs_.push_back({ domain.id, degree, constituents, contents, cache_evaluated_contents });
        return oracle_handle{ virtual_regs_.size() - 1, true };
    }
    prover_message_handle register_prover_message(std::size_t size)
    {
        update_rounds_and_direction(true);
        prover_message_sizes_.push_back(size);
        prover_messages_.emplace_back();
        message_submitted_.push_back(false);
        return prover_message_handle{ prover_message_sizes_.size() - 1 };
    }
    verifier_random_message_handle register_verifier_random_message(std::size_t size)
    {
        update_rounds_and_direction(false);
        verifier_message_sizes_.push_back(size);
        return verifier_random_message_handle{ verifier_message_sizes_.size() - 1 };
    }
    // bcs_common.tcc:482-495 with round_parameters(domain): the Merkle leaves of the current round hold cosets of |domain|
    void set_round_parameters(const field_subset<FieldT> &quotient_map_domain)
    {
        const std::size_t cur_round = num_interaction_rounds_;
        if (!round_params_.empty() && cur_round == round_params_.size() - 1) throw std::logic_error("Already set round parameters for this round");
        while (round_params_.size() < cur_round) round_params_.push_back(1);
        round_params_.push_back(quotient_map_domain.num_elements());
    }
    domain_handle get_oracle_domain(const oracle_handle &h) const { return domain_handle{ h.is_virtual ? virtual_regs_[h.id].domain : oracle_regs_[h.id].domain }; }
    std::size_t get_oracle_degree(const oracle_handle &h) const { return h.is_virtual ? virtual_regs_[h.id].degree : oracle_regs_[h.id].degree; }

    void seal_interaction_registrations()                                                    // iop.tcc:227-251, bcs_common.tcc:423-480
    {
        if (!from_prover_) throw std::logic_error("attempted to seal interaction registrations where verifier sends the last interactive message");
        num_oracles_at_end_of_round_.push_back(oracle_regs_.size());
        num_prover_messages_at_end_of_round_.push_back(prover_message_sizes_.size());
        ++num_interaction_rounds_;
        sealed_ = true;
        for (std::size_t round = 0; round < num_interaction_rounds_; ++round)
            for (auto &kv : oracles_in_round_by_domain(round)) {
                MT_info_.push_back({ round, kv.first, kv.second });
                MT_trees_.emplace_back();
                MT_roots_.emplace_back();
            }
        root_pending_.assign(MT_roots_.size(), 0);
        root_bytes_.assign(MT_roots_.size() * 32, 0);
    }
    query_position_handle register_random_query_position(const domain_handle &domain)
    {
        random_position_domains_.push_back(domain.id);
        return query_position_handle{ random_position_domains_.size() - 1, true };
    }
    query_position_handle register_deterministic_query_position(const std::vector<query_position_handle> &seeds, const position_calculator &calculator)
    {
        deterministic_positions_.emplace_back(seeds, calculator);
        return query_position_handle{ deterministic_positions_.size() - 1, false };
    }
    query_handle register_query(const oracle_handle &oracle, const query_position_handle &position)
    {
        queries_.emplace_back(oracle, position);
        return query_handle{ queries_.size() - 1 };
    }
    void seal_query_registrations() {}

    // ---- proving (iop.tcc:265-433, bcs_prover.tcc:23-98) ----
    void submit_oracle(const oracle_handle &handle, const oracle<FieldT> &contents)
    {
        const std::size_t oid = handle.id;
        if (handle.is_virtual) throw std::invalid_argument("cannot submit a virtual oracle");
        if (oracle_submitted_[oid]) throw std::invalid_argument("attempted to submit already submitted oracle");
        const std::size_t begin = num_prover_rounds_done_ == 0 ? 0 : num_oracles_at_end_of_round_[num_prover_rounds_done_ - 1];
        if (oid < begin) throw std::invalid_argument("submitting an oracle for a previous round");
        if (oid >= num_oracles_at_end_of_round_[num_prover_rounds_done_])
            throw std::invalid_argument("submitting an oracle for a future round (did you forget to call signal_prover_round_done?)");
        if (dist::local_size(domains_[oracle_regs_[oid].domain]) != contents.size()) throw std::invalid_argument("oracle evaluations don't match the domain size");
        oracles_[oid] = contents.device_contents();
        oracle_submitted_[oid] = true;
        for (std::size_t k = 0; k < collector_.produced.size(); ++k)                             // windows its transform already wrote
            if (collector_.produced[k].codeword.data() == oracles_[oid].data() && collector_.produced[k].codeword.size() == oracles_[oid].size()) {
                auto &cache = real_window_cache_[oid];
                for (auto &w : collector_.produced[k].windows) cache.push_back(std::move(w));
                collector_.produced.erase(collector_.produced.begin() + k);
                break;
            }
    }
    void submit_prover_message(const prover_message_handle &handle, const std::vector<FieldT> &contents)
    {
        if (message_submitted_[handle.id]) throw std::invalid_argument("attempted to submit already submitted prover message");
        if (prover_message_sizes_[handle.id] != contents.size()) throw std::invalid_argument("prover message submission does not match its registered size");
        prover_messages_[handle.id] = contents;
        message_submitted_[handle.id] = true;
    }
    // iop_protocol::submit_prover_index (iop.tcc:309-341) + bcs_prover::signal_index_submissions_done (bcs_prover.tcc:68-80): round 0's
    // oracles, messages and trees come from the index; only the hashchain runs
    void submit_prover_index(const bcs_prover_index<FieldT> &index)
    {
        if (num_prover_rounds_done_ != 0) throw std::invalid_argument("The IOP prover index should only be for round 0");
        const std::size_t count = num_oracles_at_end_of_round_[0];
        if (index.oracles.size() != count) throw std::invalid_argument("The IOP prover index provided the wrong number of evaluations");
        for (std::size_t oid = 0; oid < count; ++oid) submit_oracle(oracle_handle{ oid, false }, oracle<FieldT>(index.oracles[oid]));
        for (std::size_t mid = 0; mid < num_prover_messages_at_end_of_round_[0]; ++mid) submit_prover_message(prover_message_handle{ mid }, index.prover_messages[mid]);
        signal_index_submissions_done();
    }
    // bcs_indexer.tcc:17-53 when no index was given (the trees are built here), bcs_prover.tcc:68-80 otherwise
    void signal_index_submissions_done()
    {
        if (num_prover_rounds_done_ != 0) throw std::invalid_argument("Index submissions should be round 0");
        finish_round(index_ == nullptr);
    }
    std::size_t num_index_trees() const { return oracles_in_round_by_domain(0).size(); }
    bcs_prover_index<FieldT> get_prover_index()                                              // bcs_indexer::get_bcs_prover_index (bcs_indexer.tcc:80-103)
    {
        resolve_roots();
        bcs_prover_index<FieldT> idx;
        const std::size_t k = num_index_trees(), count = num_oracles_at_end_of_round_[0];
        idx.oracles.assign(oracles_.begin(), oracles_.begin() + count);
        idx.trees.assign(MT_trees_.begin(), MT_trees_.begin() + k);
        idx.roots.assign(MT_roots_.begin(), MT_roots_.begin() + k);
        idx.prover_messages.assign(prover_messages_.begin(), prover_messages_.begin() + num_prover_messages_at_end_of_round_[0]);
        return idx;
    }
    bcs_verifier_index get_verifier_index()
    {
        resolve_roots();
        bcs_verifier_index v;
        v.index_MT_roots_.assign(MT_roots_.begin(), MT_roots_.begin() + num_index_trees());
        return v;
    }
    void signal_prover_round_done() { finish_round(true); }
private:
    std::vector<char> root_pending_;            // trees whose root has not been read back yet
    std::vector<uint8_t> root_bytes_;           // 32 bytes per tree: the target of the queued read-backs
    // schedule options of the library's table (iopx_set_option / the environment, include/libiop_amd.h), looked up per round
    static bool merkle_aside() { return iopx_get_option("IOPX_MERKLE_STREAM", 1) != 0; }      // 0: every round's tree on the main stream
    static bool defer_roots() { return iopx_get_option("IOPX_DEFER_ROOTS", 1) != 0; }         // 0: every root read back at its round end
    // queues the read-back of every pending root (inside a defer window: delivered by its end); finish_pending_roots turns the bytes into digests
    void queue_pending_roots()
    {
        for (std::size_t mt = 0; mt < root_pending_.size(); ++mt)
            if (root_pending_[mt]) MT_trees_[mt].get_root_into(root_bytes_.data() + 32 * mt);
    }
    void finish_pending_roots()
    {
        for (std::size_t mt = 0; mt < root_pending_.size(); ++mt)
            if (root_pending_[mt]) { MT_roots_[mt] = hash_digest(reinterpret_cast<const char *>(root_bytes_.data() + 32 * mt), 32); root_pending_[mt] = 0; }
    }
    void resolve_roots()
    {
        bool any = false;
        for (char c : root_pending_) any = any || c;
        if (!any) return;
        check(iopx_side_stream_join());
        queue_pending_roots();                  // outside a window: immediate copies
        finish_pending_roots();
    }
    void finish_round(bool build_trees)
    {
        if (num_prover_rounds_done_ >= num_interaction_rounds_) throw std::logic_error("attempting to signal end of a round after protocol already finished");
        const std::size_t ended = num_prover_rounds_done_;
        const std::size_t begin = ended == 0 ? 0 : num_oracles_at_end_of_round_[ended - 1];
        for (std::size_t oid = begin; oid < num_oracles_at_end_of_round_[ended]; ++oid)
            if (!oracle_submitted_[oid]) throw std::logic_error("signaling end of round without submitting all oracles in the round");
        const std::size_t mbegin = ended == 0 ? 0 : num_prover_messages_at_end_of_round_[ended - 1];
        for (std::size_t mid = mbegin; mid < num_prover_messages_at_end_of_round_[ended]; ++mid)
            if (!message_submitted_[mid]) throw std::logic_error("signaling end of round without submitting all prover messages in the round");
        ++num_prover_rounds_done_;
        collector_.produced.clear();                                  // windows of transforms whose output was not submitted as an oracle of this round
        // one tree per (round, domain) over every oracle of that domain, leaves serialised by cosets (bcs_prover.tcc:36-47); the
        // reference indexes Merkle_trees_[processed_MTs_] for each domain of the round: every shipped protocol has one
        const auto mapping = oracles_in_round_by_domain(ended);
        if (mapping.size() > 1) throw std::logic_error("more than one oracle domain in a round (bcs_prover.tcc:36-47)");
        const std::size_t cs = get_round_parameters(ended);
        std::size_t num_roots = 0;
        for (auto &kv : mapping) {
            if (build_trees) {
                std::vector<device_vector<FieldT>> round_oracles;
                for (std::size_t oid : kv.second) round_oracles.push_back(oracles_[oid]);
                // ... and nothing else in the next round reads the tree, so its kernels (leaves, levels and the single-workgroup top: 5.7 ms of a
                // 2^20 proof, 0.9 ms of it latency-bound) go to the library's side stream and run beside the next round's transforms; the query
                // phase joins.  Not when distributed: the tree's collectives stay in the communicator's one stream order.  IOPX_MERKLE_STREAM=0: main stream.
                const bool deferring = defer_roots() && !blake2b_hashchain<FieldT>::absorbs_input;      // a chain that absorbs needs every root before its round's challenges
                const bool aside = deferring && merkle_aside() && !dist::ctx().active();
                if (aside) check(iopx_side_stream_begin());
                struct back_to_main { bool on; ~back_to_main() { if (on) (void)iopx_side_stream_end(); } } section{ aside };
                MT_trees_[processed_MTs_] = device_merkle_tree(round_oracles, domains_[kv.first], cs);
                // The root is needed on the host for the TRANSCRIPT only: blake2b_hashchain::absorb never mixes its input in (reference quirk F8,
                // bcs/hashing/blake2b.tcc:51-66), so every challenge is a function of the round structure and the prover may go on enqueuing the
                // next round without waiting for this tree.  The read-back is queued with the query phase's (one drain of the stream for all of
                // them); IOPX_DEFER_ROOTS=0 reads each root at its round end, as round 4 did (11 drains of about 40 us per 2^20 proof).
                if (deferring) root_pending_[processed_MTs_] = 1;
                else MT_roots_[processed_MTs_] = MT_trees_[processed_MTs_].get_root();
            } else {                                                                         // "The Merkle trees are already filled in by the preprocessor."
                MT_trees_[processed_MTs_] = index_->trees[processed_MTs_];
                MT_roots_[processed_MTs_] = index_->roots[processed_MTs_];
            }
            ++processed_MTs_;
            ++num_roots;
        }
        run_hashchain_for_round(ended, num_roots);
        // bcs_prover.tcc:52-59; the indexer's one-round protocol registers no proof of work (bcs_common.tcc:426-431)
        if (num_prover_rounds_done_ == num_interaction_rounds_ && !(is_holographic_ && num_interaction_rounds_ == 1)) {
            // The query phase (positions from the hashchain, gathers, one drain) does not depend on the answer: on one GPU it runs behind the grind's
            // first long batch.  The challenge is squeezed first, the query positions after it: the hashchain's order is the reference's either way.
            const std::string challenge = hashchain_.squeeze_root_type();
            pow_answer_ = dist::solve_pow(challenge, pow_bitlen_, [this] { extracted_ = extract_queries(); have_extracted_ = true; });   // split by candidate range over the ranks
        }
    }
public:
    std::vector<FieldT> obtain_verifier_random_message(const verifier_random_message_handle &h) const
    {
        auto it = verifier_random_messages_.find(h.id);
        if (it == verifier_random_messages_.end())
            throw std::logic_error("verifier random message not available yet (did you forget to call signal_prover_round_done?)");
        return it->second;
    }
    // iop.tcc:630-667: the oracle's evaluations in HBM; virtual oracles are evaluated from their constituents
    device_vector<FieldT> get_oracle_evaluations(const oracle_handle &h)
    {
        if (!h.is_virtual) return oracles_[h.id];
        auto it = virtual_contents_cache_.find(h.id);
        if (it != virtual_contents_cache_.end()) return it->second;
        const virtual_registration &reg = virtual_regs_[h.id];
        std::vector<device_vector<FieldT>> constituents;
        for (auto &c : reg.constituents) constituents.push_back(get_oracle_evaluations(c));
        device_vector<FieldT> result = reg.contents->evaluated_contents(constituents);
        if (reg.cache) virtual_contents_cache_[h.id] = result;
        return result;
    }
    // The oracle over a window of its domain D only (dist::window: a sub-domain in its own right) — above all the head of D, the `count`
    // positions that determine a polynomial of `count` coefficients (the ones IFFT_of_known_degree reads, fft.tcc:435-475).  A virtual
    // oracle is a pointwise map of its constituents, so its values on the window follow from the constituents' values there: |D| / count times
    // less work than get_oracle_evaluations when only the oracle's polynomial is wanted.  Same field elements as the corresponding entries of
    // get_oracle_evaluations(h).  Over a distributed domain: only on the rank that holds the window (dist::window_owner).
    bool can_restrict(const oracle_handle &h) const
    {
        if (!h.is_virtual) return true;
        const virtual_registration &reg = virtual_regs_[h.id];
        if (!reg.contents->restrictable()) return false;
        for (auto &c : reg.constituents) {
            if ((c.is_virtual ? virtual_regs_[c.id].domain : oracle_regs_[c.id].domain) != reg.domain) return false;
            if (!can_restrict(c)) return false;
        }
        return true;
    }
    // the head a later stage of the protocol will ask for (the LDT's): an earlier stage that needs a smaller head of a cached oracle evaluates
    // the larger one once instead of both
    void set_head_hint(std::size_t count) { head_hint_ = std::max(head_hint_, count); }
    // A protocol that will ask for window w of the real oracles over this domain (get_oracle_evaluations_over_window) says so while it registers: over
    // a multiplicative coset held whole the forward transform then writes w beside the codeword (dist::window_collector) instead of a strided sweep later.
    void want_window(const domain_handle &h, const dist::window &w)
    {
        const field_subset<FieldT> &D = domains_[h.id];
        if (D.type() == affine_subspace_type || D.distributed() || w.stride == 1 || (w.stride & (w.stride - 1)) || w.count * w.stride != D.num_elements()) return;
        if (collector_.domain_elements && collector_.domain_elements != D.num_elements()) return;       // one codeword domain per proof
        collector_.domain_elements = D.num_elements();
        for (auto &have : collector_.wanted) if (have == w) return;
        if (collector_.wanted.size() < 2) collector_.wanted.push_back(w);
    }
    std::size_t head_hint() const { return head_hint_; }
    std::size_t smallest_window(const oracle_handle &h) const
    {
        if (!h.is_virtual) return 1;
        const virtual_registration &reg = virtual_regs_[h.id];
        std::size_t m = reg.contents->smallest_window();
        for (auto &c : reg.constituents) m = std::max(m, smallest_window(c));
        return m;
    }
    device_vector<FieldT> get_oracle_evaluations_over_window(const oracle_handle &h, const dist::window &w)
    {
        const field_subset<FieldT> &D = domains_[h.is_virtual ? virtual_regs_[h.id].domain : oracle_regs_[h.id].domain];
        const bool indexed = !h.is_virtual && index_ && is_holographic_ && !num_oracles_at_end_of_round_.empty() && h.id < num_oracles_at_end_of_round_[0];
        auto &cache = h.is_virtual ? window_cache_[h.id] : (indexed ? index_->oracle_windows[h.id] : real_window_cache_[h.id]);
        device_vector<FieldT> result;
        for (auto &e : cache) if (dist::window_from_window<FieldT>(e.second, e.first, w, D.type(), result)) return result;
        if (!h.is_virtual) {
            result = dist::window_of<FieldT>(oracles_[h.id], D, w);
            if (D.type() != affine_subspace_type) cache.emplace_back(w, result);              // a strided gather: once per oracle and window (subspaces: a view)
            return result;
        }
        const virtual_registration &reg = virtual_regs_[h.id];
        std::vector<device_vector<FieldT>> constituents;
        for (auto &c : reg.constituents) constituents.push_back(get_oracle_evaluations_over_window(c, w));
        result = reg.contents->evaluated_contents_over(dist::window_domain(D, w), constituents);
        if (reg.cache) cache.emplace_back(w, result);
        return result;
    }
    device_vector<FieldT> get_oracle_evaluations_over_head(const oracle_handle &h, std::size_t count)
    {
        return get_oracle_evaluations_over_window(h, dist::head_window(domains_[h.is_virtual ? virtual_regs_[h.id].domain : oracle_regs_[h.id].domain], count));
    }

    // ---- transcript (bcs_prover.tcc:136-233) ----
private:
    // what both forms of the transcript are made of: per tree the sorted distinct query / leaf positions, the answers (position-major rows of
    // `width` elements) and the authentication path bytes, read back through ONE drain of the stream
    struct extracted {
        std::vector<std::vector<std::size_t>> query_positions, leaf_positions;
        std::vector<std::vector<FieldT>> flat_responses;
        std::vector<std::vector<uint8_t>> proof_bytes;
    };
    extracted extract_queries()
    {
        extracted x;
        check(iopx_side_stream_join());                                                      // the trees built beside the rounds: the gathers below read them
        std::map<std::size_t, std::size_t> random_cache, det_cache;
        std::vector<std::vector<std::size_t>> positions_by_oracle(oracle_regs_.size());
        for (auto &q : queries_) record(q.first, obtain_query_position(q.second, random_cache, det_cache), positions_by_oracle);   // registration order
        // the two small read-backs per tree are queued, not waited for one by one (iopx_defer_downloads_begin / _end)
        x.flat_responses.resize(MT_info_.size());
        x.proof_bytes.resize(MT_info_.size());
        // Distributed trees: every queried row (an answer row, an authentication-path node) has exactly one owner.  All trees' rows go into ONE zeroed
        // device buffer — the arena — each rank writes the rows it owns, and a single all-reduce completes it everywhere: one collective per proof
        // for the whole query phase instead of two per tree (22 for Aurora's eleven trees).  First the positions and the arena's layout ...
        struct plan { std::vector<std::size_t> qpos, lpos, proof_nodes; std::size_t answers_at = 0, proof_at = 0; };
        std::vector<plan> plans(MT_info_.size());
        std::size_t arena_bytes = 0;
        for (std::size_t mt = 0; mt < MT_info_.size(); ++mt) {
            const tree_info &info = MT_info_[mt];
            const std::size_t cs = get_round_parameters(info.round);
            const field_subset<FieldT> &domain = domains_[info.domain];
            const std::size_t num_leaves = domain.num_elements() / cs;
            std::vector<std::size_t> &qpos = plans[mt].qpos, &lpos = plans[mt].lpos;         // sorted and distinct, as the reference's sets are
            for (std::size_t oid : info.oracle_ids) qpos.insert(qpos.end(), positions_by_oracle[oid].begin(), positions_by_oracle[oid].end());
            std::sort(qpos.begin(), qpos.end());
            qpos.erase(std::unique(qpos.begin(), qpos.end()), qpos.end());
            for (std::size_t pos : qpos) lpos.push_back(cs == 1 ? pos : (domain.type() == affine_subspace_type ? pos / cs : pos % num_leaves));   // bcs_common.tcc:682-696
            std::sort(lpos.begin(), lpos.end());
            lpos.erase(std::unique(lpos.begin(), lpos.end()), lpos.end());
            x.flat_responses[mt].resize(qpos.size() * info.oracle_ids.size());
            if (domain.distributed() && !qpos.empty()) { plans[mt].answers_at = arena_bytes; arena_bytes += x.flat_responses[mt].size() * sizeof(FieldT); }
            if (MT_trees_[mt].distributed() && !lpos.empty()) {
                plans[mt].proof_nodes = MT_trees_[mt].distributed_proof_nodes(lpos);
                x.proof_bytes[mt].resize(32 * plans[mt].proof_nodes.size());
                plans[mt].proof_at = arena_bytes;
                arena_bytes += x.proof_bytes[mt].size();
            }
        }
        const device_array<uint8_t> arena(arena_bytes ? arena_bytes : 8);
        if (arena_bytes) arena.fill_zero();
        check(iopx_defer_downloads_begin());
        bool deferring = true;
        struct end_on_unwind { bool &on; ~end_on_unwind() { if (on) (void)iopx_defer_downloads_end(); } } guard{ deferring };
        // ... then every tree's gathers (no collective) ...
        for (std::size_t mt = 0; mt < MT_info_.size(); ++mt) {
            const tree_info &info = MT_info_[mt];
            const field_subset<FieldT> &domain = domains_[info.domain];
            const std::vector<std::size_t> &qpos = plans[mt].qpos, &lpos = plans[mt].lpos;
            if (!qpos.empty()) {                                                             // bcs_prover.tcc:187-197
                std::vector<const void *> ptrs;
                for (std::size_t oid : info.oracle_ids) ptrs.push_back(oracles_[oid].data());
                if (!domain.distributed()) {
                    check(iopx_query_responses_dev(ptrs.data(), ptrs.size(), sizeof(FieldT), domain.num_elements(), qpos.data(), qpos.size(), x.flat_responses[mt].data()));
                } else {
                    const dist::context &c = dist::ctx();
                    const std::size_t block = domain.num_elements() / c.world;
                    std::vector<uint64_t> rows, local_index;
                    for (std::size_t row = 0; row < qpos.size(); ++row) {
                        const std::size_t p = qpos[row];
                        const bool mine = domain.type() == affine_subspace_type ? p / block == c.rank : p % c.world == c.rank;
                        if (mine) { rows.push_back(row); local_index.push_back(domain.type() == affine_subspace_type ? p - c.rank * block : p / c.world); }
                    }
                    if (!rows.empty()) check(iopx_gather_rows_dev(ptrs.data(), ptrs.size(), sizeof(FieldT), local_index.data(), rows.data(), rows.size(), arena.data() + plans[mt].answers_at));
                }
            }
            if (MT_trees_[mt].distributed()) {
                if (!plans[mt].proof_nodes.empty()) MT_trees_[mt].gather_owned_proof_nodes(plans[mt].proof_nodes, arena.data() + plans[mt].proof_at);
            } else {
                x.proof_bytes[mt] = MT_trees_[mt].get_set_membership_proof_bytes(lpos);
            }
        }
        // ... one all-reduce, and the read-backs of its parts
        if (arena_bytes) {
            check(iopx_comm_all_reduce_u64_dev(dist::ctx().comm, arena.data(), arena_bytes / 8, IOPX_COMM_SUM));
            for (std::size_t mt = 0; mt < MT_info_.size(); ++mt) {
                if (domains_[MT_info_[mt].domain].distributed() && !plans[mt].qpos.empty())
                    check(iopx_memcpy_d2h_deferrable(x.flat_responses[mt].data(), arena.data() + plans[mt].answers_at, x.flat_responses[mt].size() * sizeof(FieldT)));
                if (!plans[mt].proof_nodes.empty())
                    check(iopx_memcpy_d2h_deferrable(x.proof_bytes[mt].data(), arena.data() + plans[mt].proof_at, x.proof_bytes[mt].size()));
            }
        }
        for (std::size_t mt = 0; mt < MT_info_.size(); ++mt) {
            x.query_positions.push_back(std::move(plans[mt].qpos));
            x.leaf_positions.push_back(std::move(plans[mt].lpos));
        }
        queue_pending_roots();                                                               // the roots travel with the answers and the paths
        check(iopx_defer_downloads_end());                                                   // one drain of the stream delivers every read-back queued above
        deferring = false;
        finish_pending_roots();
        return x;
    }
    extracted extracted_;
    bool have_extracted_ = false;
    // the query phase's results: computed behind the proof-of-work grind when the protocol has one, here otherwise (and again for a second transcript)
    extracted take_extracted()
    {
        if (!have_extracted_) return extract_queries();
        have_extracted_ = false;
        return std::move(extracted_);
    }
public:
    bcs_transformation_transcript<FieldT> get_transcript()
    {
        bcs_transformation_transcript<FieldT> t;
        t.prover_messages_ = prover_messages_;
        extracted x = take_extracted();
        resolve_roots();
        t.MT_roots_ = MT_roots_;
        t.query_positions_ = std::move(x.query_positions);
        t.MT_leaf_positions_ = std::move(x.leaf_positions);
        for (std::size_t mt = 0; mt < MT_info_.size(); ++mt) {
            const std::size_t width = MT_info_[mt].oracle_ids.size(), count = t.query_positions_[mt].size();
            std::vector<std::vector<FieldT>> responses(count, std::vector<FieldT>(width));
            for (std::size_t p = 0; p < count; ++p)
                for (std::size_t k = 0; k < width; ++k) responses[p][k] = x.flat_responses[mt][p * width + k];
            t.query_responses_.push_back(responses);
            t.MT_set_membership_proofs_.push_back(device_merkle_tree::digests_of(x.proof_bytes[mt]));
        }
        if (is_holographic_) {                                                               // remove_index_info_from_transcript (bcs_prover.tcc:119-134)
            t.prover_messages_.erase(t.prover_messages_.begin(), t.prover_messages_.begin() + num_prover_messages_at_end_of_round_[0]);
            t.MT_roots_.erase(t.MT_roots_.begin(), t.MT_roots_.begin() + num_index_trees());
        }
        t.proof_of_work_ = pow_answer_;
        return t;
    }
    // bcs_transformation_transcript::serialize() of get_transcript(), written straight from the read-back buffers: the structured form costs some
    // five thousand small allocations per 2^20 proof (one std::string per authentication-path digest, one vector per answer row) — 0.5 ms at the
    // end of every proof that a caller who wants the bytes (the C ABI) does not need to pay
    std::string get_transcript_bytes()
    {
        const extracted x = take_extracted();
        resolve_roots();
        std::string out;
        auto u64 = [&](uint64_t v) { out.append(reinterpret_cast<const char *>(&v), 8); };
        const std::size_t first_message = is_holographic_ ? num_prover_messages_at_end_of_round_[0] : 0, first_root = is_holographic_ ? num_index_trees() : 0;
        u64(prover_messages_.size() - first_message);
        for (std::size_t m = first_message; m < prover_messages_.size(); ++m) {
            u64(prover_messages_[m].size());
            out.append(reinterpret_cast<const char *>(prover_messages_[m].data()), prover_messages_[m].size() * sizeof(FieldT));
        }
        u64(MT_roots_.size() - first_root);
        for (std::size_t r = first_root; r < MT_roots_.size(); ++r) out.append(MT_roots_[r]);
        for (std::size_t mt = 0; mt < MT_info_.size(); ++mt) {
            u64(x.query_positions[mt].size());
            for (std::size_t p : x.query_positions[mt]) u64(p);
            u64(x.leaf_positions[mt].size());
            for (std::size_t p : x.leaf_positions[mt]) u64(p);
            u64(x.query_positions[mt].empty() ? 0 : MT_info_[mt].oracle_ids.size());
            out.append(reinterpret_cast<const char *>(x.flat_responses[mt].data()), x.flat_responses[mt].size() * sizeof(FieldT));
            u64(x.proof_bytes[mt].size() / 32);
            out.append(reinterpret_cast<const char *>(x.proof_bytes[mt].data()), x.proof_bytes[mt].size());
        }
        out.append(pow_answer_);
        return out;
    }
};

} // namespace libiop_amd
