// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's BCS Merkle tree construction with BLAKE2b and of its
// BLAKE2b hashchain.  Field elements are hashed as raw in-memory bytes (sizeof(FieldT) * count),
// exactly as libiop/bcs/hashing/blake2b.tcc:140-160 does.  Citations relative to /root/reference.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "blake2b.hpp"
#include "fri.hpp"

namespace oracle {

static const size_t DIGEST_LEN = 32;   // bcs_common.tcc:405 / blake2b.tcc:15,117 at security 128

// libiop/bcs/hashing/blake2b.cpp:28-48 — H(first || second)
static inline void two_to_one(const uint8_t *l, const uint8_t *r, uint8_t *out)
{
    uint8_t buf[2 * DIGEST_LEN];
    memcpy(buf, l, DIGEST_LEN);
    memcpy(buf + DIGEST_LEN, r, DIGEST_LEN);
    blake2b(out, DIGEST_LEN, buf, sizeof(buf));
}

// libiop/bcs/merkle_tree.tcc:92-151 (leaves) + :200-229 (inner nodes).
//   oracles[k]   : pointer to n elements of elem_bytes raw bytes each
//   additive     : position map of the default domain of that size (subspace.tcc:73-91 vs subgroup.tcc:175-197)
//   salts        : nullptr, or num_leaves * salt_bytes zk salts: leaf = H(H(slice) || salt) (blake2b.tcc:126-136)
//   nodes        : (2 L - 1) * 32 bytes, heap order, leaves at (L - 1) + i (merkle_tree.tcc:145)
static inline void merkle_build(const uint8_t *const *oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                                size_t coset_size, bool additive, const uint8_t *salts, size_t salt_bytes,
                                uint8_t *nodes)
{
    const size_t L = n / coset_size;
    if (L < 2 || (L & (L - 1))) throw std::invalid_argument("Merkle tree size must be a power of two, and at least 2.");
    std::vector<uint8_t> slice(num_oracles * coset_size * elem_bytes);
    for (size_t i = 0; i < L; ++i) {
        for (size_t j = 0; j < coset_size; ++j) {
            const size_t pos = position_by_coset_indices(additive, n, i, j, coset_size);
            for (size_t k = 0; k < num_oracles; ++k) {
                // slice[j + k * coset_size] = oracle_k[pos]   (merkle_tree.tcc:127-134)
                memcpy(&slice[(j + k * coset_size) * elem_bytes], oracles[k] + pos * elem_bytes, elem_bytes);
            }
        }
        uint8_t *leaf = nodes + ((L - 1) + i) * DIGEST_LEN;
        blake2b(leaf, DIGEST_LEN, slice.data(), slice.size());
        if (salts) {
            std::vector<uint8_t> buf(DIGEST_LEN + salt_bytes);
            memcpy(buf.data(), leaf, DIGEST_LEN);
            memcpy(buf.data() + DIGEST_LEN, salts + i * salt_bytes, salt_bytes);
            blake2b(leaf, DIGEST_LEN, buf.data(), buf.size());
        }
    }
    // merkle_tree.tcc:200-229: levels from n = (L-1)/2 down to 0, node j = H(node[2j+1] || node[2j+2])
    size_t lvl = (L - 1) / 2;
    while (true) {
        for (size_t j = lvl; j <= 2 * lvl; ++j) {
            two_to_one(nodes + (2 * j + 1) * DIGEST_LEN, nodes + (2 * j + 2) * DIGEST_LEN, nodes + j * DIGEST_LEN);
        }
        if (lvl > 0) lvl /= 2; else break;
    }
}

// libiop/bcs/hashing/blake2b.tcc:10-110 — the hashchain, including the reference's behaviour that
// absorb() hashes only the first digest_len bytes of state||input (:56-60).
struct blake2b_hashchain {
    uint8_t state[DIGEST_LEN];
    uint64_t squeeze_index;

    blake2b_hashchain() : squeeze_index(0) { memset(state, ' ', DIGEST_LEN); }              // :17

    void absorb_digest(const uint8_t *digest /* DIGEST_LEN bytes, ignored by the hash */)
    {
        uint8_t buf[2 * DIGEST_LEN];
        memcpy(buf, state, DIGEST_LEN);
        memcpy(buf + DIGEST_LEN, digest, DIGEST_LEN);
        uint8_t out[DIGEST_LEN];
        blake2b(out, DIGEST_LEN, buf, DIGEST_LEN);                                          // inlen = digest_len (:60)
        memcpy(state, out, DIGEST_LEN);
    }

    // :76-86 + :231-257 (binary fields, :162-185): element i of this squeeze =
    // keyed BLAKE2b(msg = state || index (8 B LE), key = i (8 B LE), outlen = elem_bytes), written raw.
    void squeeze_binary_field(size_t num_elements, size_t elem_bytes, uint8_t *out)
    {
        ++squeeze_index;
        uint8_t msg[DIGEST_LEN + 8];
        memcpy(msg, state, DIGEST_LEN);
        memcpy(msg + DIGEST_LEN, &squeeze_index, 8);
        for (uint64_t i = 0; i < num_elements; ++i) {
            blake2b(out + i * elem_bytes, elem_bytes, msg, sizeof(msg), (const uint8_t *)&i, 8);
        }
    }

    // :88-105 + blake2b.cpp:50-74: one squeeze index per position, result mod range (power of two)
    std::vector<size_t> squeeze_query_positions(size_t num_positions, size_t range)
    {
        if (range & (range - 1)) throw std::invalid_argument("upper_bound must be a power of two.");
        std::vector<size_t> out;
        for (size_t i = 0; i < num_positions; ++i) {
            ++squeeze_index;
            uint64_t r;
            blake2b((uint8_t *)&r, 8, state, DIGEST_LEN, (const uint8_t *)&squeeze_index, 8);
            out.push_back((size_t)(r % range));
        }
        return out;
    }
};

// merkle_tree::get_set_membership_proof (merkle_tree.tcc:242-336): indices into the heap-ordered node array of the
// auxiliary hashes, in the order the reference emits them; positions are leaf indices (any order, duplicates allowed)
static inline std::vector<size_t> membership_proof_node_indices(size_t num_leaves, const std::vector<size_t> &positions)
{
    std::vector<size_t> out;
    if (positions.empty()) return out;
    std::vector<size_t> S = positions;
    std::sort(S.begin(), S.end());
    S.erase(std::unique(S.begin(), S.end()), S.end());
    for (size_t pos : S) if (pos >= num_leaves) throw std::invalid_argument("All positions must be between 0 and num_leaves-1.");
    for (size_t &pos : S) pos += num_leaves - 1;
    while (true) {
        if (S.size() == 1 && S[0] == 0) break;
        std::vector<size_t> new_S;
        size_t i = 0;
        while (i < S.size()) {
            const size_t it_pos = S[i];
            size_t next = i + 1;
            new_S.push_back((it_pos - 1) / 2);
            if ((it_pos & 1) == 0) out.push_back(it_pos - 1);                   // right node: left sibling is auxiliary
            else if (next == S.size() || S[next] != it_pos + 1) out.push_back(it_pos + 1);   // a) right sibling not in S
            else ++next;                                                        // b) right sibling in S: skip it
            i = next;
        }
        S.swap(new_S);
    }
    return out;
}

// merkle_tree::validate_set_membership_proof (merkle_tree.tcc:338-483) for non-zk BLAKE2b trees, given the leaf HASHES
// of the queried positions (the caller hashes the leaf contents); positions sorted ascending, unique
static inline bool membership_proof_validate(const uint8_t *root, size_t num_leaves, const std::vector<size_t> &positions,
                                             const std::vector<std::vector<uint8_t>> &leaf_hashes, const std::vector<std::vector<uint8_t>> &aux)
{
    typedef std::pair<size_t, std::vector<uint8_t>> pd;
    std::vector<pd> S;
    for (size_t i = 0; i < positions.size(); ++i) S.push_back(pd(positions[i] + num_leaves - 1, leaf_hashes[i]));
    size_t a = 0;
    while (true) {
        if (S.size() == 1 && S[0].first == 0) break;
        std::vector<pd> new_S;
        size_t i = 0;
        while (i < S.size()) {
            const size_t it_pos = S[i].first;
            size_t next = i + 1;
            std::vector<uint8_t> l, r;
            if ((it_pos & 1) == 0) { if (a >= aux.size()) return false; l = aux[a++]; r = S[i].second; }
            else {
                l = S[i].second;
                if (next == S.size() || S[next].first != it_pos + 1) { if (a >= aux.size()) return false; r = aux[a++]; }
                else { r = S[next].second; ++next; }
            }
            std::vector<uint8_t> h(DIGEST_LEN);
            two_to_one(l.data(), r.data(), h.data());
            new_S.push_back(pd((it_pos - 1) / 2, h));
            i = next;
        }
        S.swap(new_S);
    }
    if (a != aux.size()) throw std::logic_error("Validation did not consume the entire proof.");
    return memcmp(S[0].second.data(), root, DIGEST_LEN) == 0;
}

// merkle_tree::count_hashes_to_verify_set_membership_proof (merkle_tree.tcc:485-515)
static inline size_t count_hashes_to_verify_set_membership_proof(size_t num_leaves, std::vector<size_t> positions)
{
    size_t total = 0, depth = 0;
    while (((size_t)1 << depth) < num_leaves) ++depth;
    std::sort(positions.begin(), positions.end());
    for (size_t d = depth; d > 0; --d) {
        std::vector<size_t> next;
        for (size_t p : positions) if (next.empty() || next.back() != p / 2) next.push_back(p / 2);
        total += next.size();
        positions = next;
    }
    return total;
}

} // namespace oracle
