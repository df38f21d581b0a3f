// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's BCS Merkle tree construction with BLAKE2b and of its
// BLAKE2b hashchain.  Field elements are hashed as raw in-memory bytes (sizeof(FieldT) * count),
// exactly as libiop/bcs/hashing/blake2b.tcc:140-160 does.  Citations relative to /root/reference.
#pragma once
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

} // namespace oracle
