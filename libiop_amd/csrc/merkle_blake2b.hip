// BCS Merkle tree with BLAKE2b-256 leaves and nodes on gfx950.
//
// Replaces merkle_tree::construct_with_leaves_serialized_by_cosets + compute_inner_nodes
// (libiop/bcs/merkle_tree.tcc:92-151, 200-229) with blake2b_leafhash / blake2b_two_to_one_hash
// (libiop/bcs/hashing/blake2b.tcc:120-160, blake2b.cpp:28-48).  BLAKE2b follows RFC 7693; one lane
// hashes one leaf (or one inner node), the message words are gathered straight from the oracle
// buffers in the reference's serialisation order: oracle-major, then position within the coset
// (merkle_tree.tcc:127-134).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>
#include "runtime.h"

namespace iopx {

__device__ static const uint64_t B2B_IV[8] = {
    0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
    0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };

// 64-bit rotations on the 32-bit halves: by 32 a swap, otherwise two v_alignbit_b32 (the generic shift form compiles to
// five instructions)
template<int N>
__device__ __forceinline__ uint64_t rotr64(uint64_t x)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (N == 32) return ((uint64_t)lo << 32) | hi;
    if (N < 32) return ((uint64_t)__builtin_amdgcn_alignbit(lo, hi, N) << 32) | __builtin_amdgcn_alignbit(hi, lo, N);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, N - 32) << 32) | __builtin_amdgcn_alignbit(lo, hi, N - 32);
}

#define B2B_G(a, b, c, d, x, y)                                   \
    do {                                                          \
        v[a] = v[a] + v[b] + (x); v[d] = rotr64<32>(v[d] ^ v[a]); \
        v[c] = v[c] + v[d];       v[b] = rotr64<24>(v[b] ^ v[c]); \
        v[a] = v[a] + v[b] + (y); v[d] = rotr64<16>(v[d] ^ v[a]); \
        v[c] = v[c] + v[d];       v[b] = rotr64<63>(v[b] ^ v[c]); \
    } while (0)

#define B2B_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    do {                                                                                \
        B2B_G(0, 4,  8, 12, m[s0],  m[s1]);                                             \
        B2B_G(1, 5,  9, 13, m[s2],  m[s3]);                                             \
        B2B_G(2, 6, 10, 14, m[s4],  m[s5]);                                             \
        B2B_G(3, 7, 11, 15, m[s6],  m[s7]);                                             \
        B2B_G(0, 5, 10, 15, m[s8],  m[s9]);                                             \
        B2B_G(1, 6, 11, 12, m[s10], m[s11]);                                            \
        B2B_G(2, 7,  8, 13, m[s12], m[s13]);                                            \
        B2B_G(3, 4,  9, 14, m[s14], m[s15]);                                            \
    } while (0)

// one compression: h <- F(h, m, t, last)    (RFC 7693 section 3.2; message length < 2^64)
__device__ __forceinline__ void b2b_compress(uint64_t (&h)[8], const uint64_t (&m)[16], uint64_t t, bool last)
{
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    B2B_ROUND( 0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15);
    B2B_ROUND(14, 10,  4,  8,  9, 15, 13,  6,  1, 12,  0,  2, 11,  7,  5,  3);
    B2B_ROUND(11,  8, 12,  0,  5,  2, 15, 13, 10, 14,  3,  6,  7,  1,  9,  4);
    B2B_ROUND( 7,  9,  3,  1, 13, 12, 11, 14,  2,  6,  5, 10,  4,  0, 15,  8);
    B2B_ROUND( 9,  0,  5,  7,  2,  4, 10, 15, 14,  1, 11, 12,  6,  8,  3, 13);
    B2B_ROUND( 2, 12,  6, 10,  0, 11,  8,  3,  4, 13,  7,  5, 15, 14,  1,  9);
    B2B_ROUND(12,  5,  1, 15, 14, 13,  4, 10,  0,  7,  6,  3,  9,  2,  8, 11);
    B2B_ROUND(13, 11,  7, 14, 12,  1,  3,  9,  5,  0, 15,  4,  8,  6,  2, 10);
    B2B_ROUND( 6, 15, 14,  9, 11,  3,  0,  8, 12,  2, 13,  7,  1,  4, 10,  5);
    B2B_ROUND(10,  2,  8,  4,  7,  6,  1,  5, 15, 11,  9, 14,  3, 12, 13,  0);
    B2B_ROUND( 0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15);
    B2B_ROUND(14, 10,  4,  8,  9, 15, 13,  6,  1, 12,  0,  2, 11,  7,  5,  3);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

__device__ __forceinline__ void b2b_init(uint64_t (&h)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = B2B_IV[i];
    h[0] ^= 0x01010000ull ^ 32ull;         // unkeyed, 32-byte digest
}

#define LEAF_INLINE_ORACLES 16
struct LeafParams {
    const uint64_t *const *oracles;     // device array of num_oracles device pointers, or null: then the pointers are inline_oracles
    const uint64_t *inline_oracles[LEAF_INLINE_ORACLES];    // up to 16 oracles travel in the argument block (no upload launch per tree)
    const uint64_t *salts;              // nullptr or num_leaves * salt_words words
    uint64_t *nodes;                    // (2 L - 1) * 4 words
    size_t num_oracles, elem_words, n, coset_size, num_leaves;
    size_t salt_bytes;
    int additive;
};

__global__ void k_merkle_leaves(LeafParams p)
{
    const size_t words_total = p.num_oracles * p.coset_size * p.elem_words;
    const size_t bytes_total = words_total * 8;
    const uint64_t *const *oracles = p.oracles ? p.oracles : p.inline_oracles;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.num_leaves; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t h[8];
        b2b_init(h);
        size_t done = 0;                                // words consumed
        // the serialisation cursor (oracle k, position j in the coset, word ww of the element) is the same for every lane: it
        // advances by counting, never by dividing (a 64-bit division costs about as much as a BLAKE2b round)
        uint32_t k = 0, j = 0, ww = 0;
        const uint64_t *cur = oracles[0];
        // position_by_coset_indices: subspace.tcc:86-91 / subgroup.tcc:191-197
        size_t pos = p.additive ? i * p.coset_size : i;
        const size_t pos_step = p.additive ? 1 : p.num_leaves;
        while (true) {
            uint64_t m[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                uint64_t val = 0;
                if (k < p.num_oracles) {
                    val = cur[pos * p.elem_words + ww];
                    if (++ww == p.elem_words) {
                        ww = 0;
                        pos += pos_step;
                        if (++j == p.coset_size) {
                            j = 0;
                            pos = p.additive ? i * p.coset_size : i;
                            if (++k < p.num_oracles) cur = oracles[k];
                        }
                    }
                }
                m[w] = val;
            }
            const bool last = (done + 16 >= words_total);
            const uint64_t t = last ? bytes_total : (done + 16) * 8;
            b2b_compress(h, m, t, last);
            if (last) break;
            done += 16;
        }
        if (p.salts) {
            // zk leaf: H(H(slice) || salt)   (blake2b.tcc:126-136); 32 + salt_bytes <= 128
            uint64_t m[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) m[w] = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) m[w] = h[w];
            const unsigned char *sp = (const unsigned char *)p.salts + i * p.salt_bytes;
            for (size_t b = 0; b < p.salt_bytes; ++b) m[4 + (b >> 3)] |= (uint64_t)sp[b] << (8 * (b & 7));
            b2b_init(h);
            b2b_compress(h, m, 32 + p.salt_bytes, true);
        }
        uint64_t *out = p.nodes + 4 * (p.num_leaves - 1 + i);
        out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
    }
}

// The provers' own leaf shapes — NO <= 4 oracles of 24-byte elements, cosets of CS = 2 or 4 positions, no salts — with the
// serialisation order fixed at compile time: over a subspace a leaf's slice of an oracle is 24 CS contiguous bytes, 16-byte aligned, fetched as
// 16-byte loads instead of 3 CS strided 8-byte ones; in both cases the cursor arithmetic of the general kernel is gone.  Same bytes into the same
// compressions (k_merkle_leaves 3.8 -> 2.8 ms per Aurora 2^20 proof; other shapes — three oracles, cosets of one or eight, salted leaves — take the general kernel).
typedef uint64_t b2b_u64x2 __attribute__((vector_size(16)));

// ADD = false: cosets of a multiplicative domain, position j of leaf i at i + j L (subgroup.tcc:191-197): 24-byte elements on their own, 8-byte loads.
template<int NO, int CS, bool ADD>
__global__ void __launch_bounds__(256) k_merkle_leaves_sub24(LeafParams p)
{
    constexpr int W = NO * CS * 3, BLOCKS = (W + 15) / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.num_leaves; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t w[BLOCKS * 16];
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            if (ADD) {
                const b2b_u64x2 *src = (const b2b_u64x2 *)(p.inline_oracles[k] + (size_t)3 * CS * i);
#pragma unroll
                for (int q = 0; q < (CS * 3) / 2; ++q) {
                    const b2b_u64x2 t = src[q];
                    w[k * CS * 3 + 2 * q] = t[0];
                    w[k * CS * 3 + 2 * q + 1] = t[1];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CS; ++j) {
                    const uint64_t *src = p.inline_oracles[k] + 3 * (i + (size_t)j * p.num_leaves);
                    w[(k * CS + j) * 3] = src[0];
                    w[(k * CS + j) * 3 + 1] = src[1];
                    w[(k * CS + j) * 3 + 2] = src[2];
                }
            }
        }
#pragma unroll
        for (int q = W; q < BLOCKS * 16; ++q) w[q] = 0;
        uint64_t h[8];
        b2b_init(h);
#pragma unroll
        for (int b = 0; b < BLOCKS; ++b) {
            uint64_t m[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = w[16 * b + q];
            b2b_compress(h, m, b + 1 == BLOCKS ? (uint64_t)W * 8 : (uint64_t)(b + 1) * 128, b + 1 == BLOCKS);
        }
        b2b_u64x2 *out = (b2b_u64x2 *)(p.nodes + 4 * (p.num_leaves - 1 + i));
        // node (L - 1 + i) starts 32 (L - 1 + i) bytes into a 16-byte aligned array
        b2b_u64x2 lo = { h[0], h[1] }, hi = { h[2], h[3] };
        out[0] = lo;
        out[1] = hi;
    }
}

template<int NO>
static bool launch_leaves_sub24(const LeafParams &p, size_t coset_size, unsigned grid)
{
    if (coset_size == 2 && p.additive) hipLaunchKernelGGL((k_merkle_leaves_sub24<NO, 2, true>), dim3(grid), dim3(256), 0, stream(), p);
    else if (coset_size == 4 && p.additive) hipLaunchKernelGGL((k_merkle_leaves_sub24<NO, 4, true>), dim3(grid), dim3(256), 0, stream(), p);
    else if (coset_size == 2) hipLaunchKernelGGL((k_merkle_leaves_sub24<NO, 2, false>), dim3(grid), dim3(256), 0, stream(), p);
    else if (coset_size == 4) hipLaunchKernelGGL((k_merkle_leaves_sub24<NO, 4, false>), dim3(grid), dim3(256), 0, stream(), p);
    else return false;
    return true;
}

// VEC: the node array is 16-byte aligned (every array the library allocates is): four 16-byte loads, two 16-byte stores
template<bool VEC>
__device__ __forceinline__ void node_hash(uint64_t *nodes, size_t j)
{
    uint64_t h[8], m[16];
    b2b_init(h);
    const uint64_t *l = nodes + 4 * (2 * j + 1);        // children are adjacent: 64 contiguous bytes
    if (VEC) {
        const b2b_u64x2 *lv = (const b2b_u64x2 *)l;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const b2b_u64x2 t = lv[w]; m[2 * w] = t[0]; m[2 * w + 1] = t[1]; }
    } else {
#pragma unroll
        for (int w = 0; w < 8; ++w) m[w] = l[w];
    }
#pragma unroll
    for (int w = 8; w < 16; ++w) m[w] = 0;
    b2b_compress(h, m, 64, true);
    uint64_t *out = nodes + 4 * j;
    if (VEC) {
        b2b_u64x2 lo = { h[0], h[1] }, hi = { h[2], h[3] };
        ((b2b_u64x2 *)out)[0] = lo;
        ((b2b_u64x2 *)out)[1] = hi;
    } else {
        out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
    }
}

// one tree level: nodes first .. first + count - 1      (merkle_tree.tcc:208-218)
template<bool VEC>
__global__ void k_merkle_level(uint64_t *nodes, size_t first, size_t count)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x) {
        node_hash<VEC>(nodes, first + j);
    }
}

// the top of the tree in one workgroup: levels of `count`, count/2, ..., 1 nodes (one node per thread at the widest level: the kernel sits
// between the last wide level and the root's read-back, on the proof's critical path)
template<bool VEC>
__global__ void k_merkle_top(uint64_t *nodes, size_t count)
{
    for (size_t c = count; c >= 1; c >>= 1) {
        for (size_t j = threadIdx.x; j < c; j += blockDim.x) node_hash<VEC>(nodes, (c - 1) + j);
        __threadfence_block();
        __syncthreads();
    }
}

// Transcript extraction: only the O(queries * log L) digests and the queried codeword entries leave the device.
// out[i] = 32-byte node idx[i]
__global__ void k_gather_nodes(uint64_t *out, const uint64_t *nodes, const uint64_t *idx, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * count; i += (size_t)gridDim.x * blockDim.x) {
        out[i] = nodes[4 * idx[i >> 2] + (i & 3)];
    }
}

// out[(p * num_oracles + k) * elem_words + w] = oracle_k[pos[p]] word w.  Up to 16 oracles and 256 positions (every tree of the shipped provers: 27 - 36
// query repetitions of cosets of 2 - 4 positions) travel in the argument block; larger requests through device arrays.
#define GATHER_INLINE_ORACLES 16
#define GATHER_INLINE_POSITIONS 256
struct GatherInline {
    const uint64_t *oracles[GATHER_INLINE_ORACLES];
    uint64_t pos[GATHER_INLINE_POSITIONS];
};
__global__ void k_gather_responses_inline(uint64_t *out, GatherInline g, size_t num_oracles, size_t elem_words, size_t count)
{
    const size_t total = count * num_oracles * elem_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w = i % elem_words, k = (i / elem_words) % num_oracles, q = i / (elem_words * num_oracles);
        out[i] = g.oracles[k][g.pos[q] * elem_words + w];
    }
}

__global__ void k_gather_responses(uint64_t *out, const uint64_t *const *oracles, const uint64_t *pos, size_t num_oracles,
                                   size_t elem_words, size_t count)
{
    const size_t total = count * num_oracles * elem_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w = i % elem_words, k = (i / elem_words) % num_oracles, q = i / (elem_words * num_oracles);
        out[i] = oracles[k][pos[q] * elem_words + w];
    }
}

// Proof-of-work grind, binary digests (libiop/bcs/pow.tcc:86-103,111-119,143-162).  Candidate 0 is the challenge itself,
// candidate k >= 1 the challenge with its last 8-byte word replaced by k - 1; a candidate passes when the last word of
// H(challenge || candidate) has its low `bitlen` bits zero.  One lane per candidate; the smallest passing index wins.
struct PowChallenge {
    uint64_t w[4];
};

__global__ void k_pow_blake2b(PowChallenge c, uint64_t first, uint64_t count, uint64_t mask, unsigned long long *best)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t idx = first + g;
        // the grid walks the candidates in increasing order, about one resident wave of threads at a time: once a hit is known, everything
        // above it is irrelevant (the answer is the FIRST hit, pow.tcc:86-112), so the launch ends soon after the hit instead of with the batch
        if (idx > *(volatile unsigned long long *)best) break;
        uint64_t h[8], m[16];
        b2b_init(h);
        m[0] = c.w[0]; m[1] = c.w[1]; m[2] = c.w[2]; m[3] = c.w[3];
        m[4] = c.w[0]; m[5] = c.w[1]; m[6] = c.w[2];
        m[7] = idx == 0 ? c.w[3] : idx - 1;
#pragma unroll
        for (int w = 8; w < 16; ++w) m[w] = 0;
        b2b_compress(h, m, 64, true);
        if ((h[3] & mask) == 0) atomicMin(best, (unsigned long long)idx);
    }
}

} // namespace iopx

using namespace iopx;

extern "C" {

static int merkle_blake2b_impl(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                               size_t coset_size, int domain_type, const uint8_t *d_salts, size_t salt_bytes,
                               uint8_t *d_nodes, bool leaves_only)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_oracles || num_oracles == 0 || !d_nodes) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (elem_bytes == 0 || (elem_bytes & 7)) return fail(IOPX_ERR_INVALID_ARGUMENT, "element size %zu is not a multiple of 8 bytes", elem_bytes);
    if (domain_type != IOPX_DOMAIN_ADDITIVE && domain_type != IOPX_DOMAIN_MULTIPLICATIVE)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "unsupported domain type %d", domain_type);
    if (coset_size == 0 || n % coset_size) return fail(IOPX_ERR_LOGIC, "Attempting to construct a Merkle tree with a constituent vector of wrong size");
    const size_t L = n / coset_size;
    if (L < 2 || (L & (L - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "Merkle tree size must be a power of two, and at least 2.");
    if (d_salts && salt_bytes > 96) return fail(IOPX_ERR_INVALID_ARGUMENT, "zk salt of %zu bytes does not fit one BLAKE2b block", salt_bytes);

    TmpBuf dptrs;
    if (num_oracles > LEAF_INLINE_ORACLES) {
        if ((rc = dptrs.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
        { int urc_ = upload(dptrs.p, d_oracles, num_oracles * sizeof(void *)); if (urc_ != IOPX_OK) return urc_; }
    }

    LeafParams p;
    memset(&p, 0, sizeof(p));
    p.oracles = (const uint64_t *const *)dptrs.p;                              // null when the pointers fit the argument block
    if (num_oracles <= LEAF_INLINE_ORACLES) for (size_t k = 0; k < num_oracles; ++k) p.inline_oracles[k] = (const uint64_t *)d_oracles[k];
    p.salts = (const uint64_t *)d_salts;
    p.nodes = (uint64_t *)d_nodes;
    p.num_oracles = num_oracles; p.elem_words = elem_bytes / 8; p.n = n; p.coset_size = coset_size; p.num_leaves = L;
    p.salt_bytes = salt_bytes;
    p.additive = (domain_type == IOPX_DOMAIN_ADDITIVE);
    size_t grid = (L + 255) / 256;
    if (grid > 65536) grid = 65536;
    {
        // the fixed-shape kernel where it applies
        const bool fixed_ok = true;
        bool fixed = fixed_ok && elem_bytes == 24 && !d_salts && num_oracles <= 4 && (coset_size == 2 || coset_size == 4) && ((uintptr_t)d_nodes & 15) == 0;
        for (size_t k = 0; fixed && p.additive && k < num_oracles; ++k) fixed = ((uintptr_t)d_oracles[k] & 15) == 0;
        // profile names per shape (oracles x coset size), as tools/make_traffic_json.py derives them from the kernel symbols
        static const char *const shape_names[4][2] = { { "k_merkle_leaves_1x2", "k_merkle_leaves_1x4" }, { "k_merkle_leaves_2x2", "k_merkle_leaves_2x4" },
                                                      { "k_merkle_leaves_3x2", "k_merkle_leaves_3x4" }, { "k_merkle_leaves_4x2", "k_merkle_leaves_4x4" } };
        ProfScope ps_(fixed ? shape_names[num_oracles - 1][coset_size == 4] : "k_merkle_leaves", num_oracles * n * elem_bytes + L * 32);
        if (fixed) {
            switch (num_oracles) {
                case 1: fixed = launch_leaves_sub24<1>(p, coset_size, (unsigned)grid); break;
                case 2: fixed = launch_leaves_sub24<2>(p, coset_size, (unsigned)grid); break;
                case 3: fixed = launch_leaves_sub24<3>(p, coset_size, (unsigned)grid); break;
                default: fixed = launch_leaves_sub24<4>(p, coset_size, (unsigned)grid); break;
            }
        }
        if (!fixed) hipLaunchKernelGGL(k_merkle_leaves, dim3((unsigned)grid), dim3(256), 0, stream(), p);
    }

    if (leaves_only) return IOPX_OK;
    return iopx_merkle_inner_blake2b_dev(d_nodes, L);   // the pointer table is released in stream order
}

int iopx_merkle_blake2b_dev(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                            size_t coset_size, int domain_type, const uint8_t *d_salts, size_t salt_bytes,
                            uint8_t *d_nodes)
{
    return merkle_blake2b_impl(d_oracles, num_oracles, elem_bytes, n, coset_size, domain_type, d_salts, salt_bytes, d_nodes, false);
}

// the leaf digests only (nodes[L-1 .. 2L-2]); the inner nodes are left untouched
int iopx_merkle_leaves_blake2b_dev(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                                   size_t coset_size, int domain_type, const uint8_t *d_salts, size_t salt_bytes,
                                   uint8_t *d_nodes)
{
    return merkle_blake2b_impl(d_oracles, num_oracles, elem_bytes, n, coset_size, domain_type, d_salts, salt_bytes, d_nodes, true);
}

// merkle_tree::compute_inner_nodes (merkle_tree.tcc:200-229) on a node array whose leaf digests are already in place
int iopx_merkle_inner_blake2b_dev(uint8_t *d_nodes, size_t num_leaves)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_nodes) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const size_t L = num_leaves;
    if (L < 2 || (L & (L - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "Merkle tree size must be a power of two, and at least 2.");
    // inner levels: L/2, L/4, ... nodes; the last levels (<= 256 nodes: one wavefront per SIMD of one CU, a level then costs one compression's
    // latency) in one workgroup.  Wider levels take more time inside that workgroup (1024 nodes: four wavefronts per SIMD in turn) than as a launch of their own
    const bool vec = ((uintptr_t)d_nodes & 15) == 0;
    size_t count = L / 2;
    while (count > 256) {
        size_t g = (count + 63) / 64;                   // 64 nodes per workgroup at the narrow levels: 1024 nodes spread over 16 CUs
        if (g > 65536) g = 65536;
        { ProfScope ps_("k_merkle_level", count * 96);
          const unsigned tb = count >= 65536 ? 256 : 64;
          if (tb == 256) g = (count + 255) / 256;
          if (g > 65536) g = 65536;
          if (vec) hipLaunchKernelGGL(k_merkle_level<true>, dim3((unsigned)g), dim3(tb), 0, stream(), (uint64_t *)d_nodes, count - 1, count);
          else hipLaunchKernelGGL(k_merkle_level<false>, dim3((unsigned)g), dim3(tb), 0, stream(), (uint64_t *)d_nodes, count - 1, count); }
        count >>= 1;
    }
    { ProfScope ps_("k_merkle_top", count * 96);
      const dim3 tb(count >= 1024 ? 1024 : (count >= 256 ? 256 : 64));
      if (vec) hipLaunchKernelGGL(k_merkle_top<true>, dim3(1), tb, 0, stream(), (uint64_t *)d_nodes, count);
      else hipLaunchKernelGGL(k_merkle_top<false>, dim3(1), tb, 0, stream(), (uint64_t *)d_nodes, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_merkle_blake2b(const void *const *oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                        size_t coset_size, int domain_type, const uint8_t *salts, size_t salt_bytes,
                        uint8_t *nodes)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!oracles || num_oracles == 0 || !nodes) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (coset_size == 0 || n % coset_size) return fail(IOPX_ERR_LOGIC, "Attempting to construct a Merkle tree with a constituent vector of wrong size");
    const size_t L = n / coset_size;
    if (L < 2 || (L & (L - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "Merkle tree size must be a power of two, and at least 2.");
    std::vector<std::unique_ptr<DevBuf>> bufs;
    std::vector<const void *> dptrs;
    for (size_t k = 0; k < num_oracles; ++k) {
        bufs.emplace_back(new DevBuf());
        if ((rc = bufs.back()->alloc(n * elem_bytes)) != IOPX_OK) return rc;
        IOPX_HIP(copy_h2d(bufs.back()->p, oracles[k], n * elem_bytes, stream()));
        dptrs.push_back(bufs.back()->p);
    }
    DevBuf dsalt, dnodes;
    if (salts) {
        if ((rc = dsalt.alloc(L * salt_bytes)) != IOPX_OK) return rc;
        IOPX_HIP(copy_h2d(dsalt.p, salts, L * salt_bytes, stream()));
    }
    if ((rc = dnodes.alloc((2 * L - 1) * 32)) != IOPX_OK) return rc;
    rc = iopx_merkle_blake2b_dev(dptrs.data(), num_oracles, elem_bytes, n, coset_size, domain_type,
                                 salts ? (const uint8_t *)dsalt.p : nullptr, salt_bytes, (uint8_t *)dnodes.p);
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(nodes, dnodes.p, (2 * L - 1) * 32, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

// The search in two halves, so that the caller can put host work (and further launches) behind a long batch: _begin enqueues the batch on the
// library's stream and returns, _end reads the result back (one pending search per host thread's library state).
static thread_local TmpBuf g_pow_best;
static thread_local bool g_pow_pending = false;

int iopx_pow_search_blake2b_begin(const uint8_t *challenge, size_t pow_bitlen, uint64_t first, uint64_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!challenge) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (pow_bitlen > 30) return fail(IOPX_ERR_INVALID_ARGUMENT, "pow_bitlen %zu: the reference's `1 << pow_bitlen` is an int shift", pow_bitlen);
    if (g_pow_pending) return fail(IOPX_ERR_LOGIC, "a proof-of-work search is already pending");
    PowChallenge c;
    memcpy(c.w, challenge, 32);
    const uint64_t mask = ((uint64_t)1 << pow_bitlen) - 1;
    if ((rc = g_pow_best.alloc(8)) != IOPX_OK) return rc;
    const unsigned long long none = ~0ull;
    if ((rc = upload(g_pow_best.p, &none, 8)) != IOPX_OK) { g_pow_best.release(); return rc; }
    if (count) {
        ProfScope ps_("k_pow_blake2b");
        hipLaunchKernelGGL(k_pow_blake2b, dim3((unsigned)((count + 255) / 256 > 1536 ? 1536 : (count + 255) / 256)), dim3(256), 0, stream(),   // <= one resident set of workgroups (256 CUs x 6): the grid advances through the candidates together
                           c, first, count, mask, (unsigned long long *)g_pow_best.p);
    }
    g_pow_pending = true;
    return IOPX_OK;
}

int iopx_pow_search_blake2b_end(uint64_t *found)
{
    if (!found) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (!g_pow_pending) return fail(IOPX_ERR_LOGIC, "no proof-of-work search is pending");
    g_pow_pending = false;
    unsigned long long hit = ~0ull;
    const int rc = download(&hit, g_pow_best.p, 8);
    g_pow_best.release();
    if (rc != IOPX_OK) return rc;
    *found = hit;
    return IOPX_OK;
}

// Candidates in the reference's order (pow.tcc:86-112): index 0 is the challenge itself, index i >= 1 the challenge with its last
// 8-byte word set to i - 1.  Searches [first, first + count) and reports the smallest passing index, or ~0.
int iopx_pow_search_blake2b(const uint8_t *challenge, size_t pow_bitlen, uint64_t first, uint64_t count, uint64_t *found)
{
    if (!found) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = iopx_pow_search_blake2b_begin(challenge, pow_bitlen, first, count);
    if (rc != IOPX_OK) return rc;
    return iopx_pow_search_blake2b_end(found);
}

// the 32-byte answer of candidate `index` (see iopx_pow_search_blake2b)
int iopx_pow_candidate_blake2b(const uint8_t *challenge, uint64_t index, uint8_t *pow)
{
    if (!challenge || !pow) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(pow, challenge, 32);
    if (index != 0) { const uint64_t v = index - 1; memcpy(pow + 24, &v, 8); }
    return IOPX_OK;
}

int iopx_pow_solve_blake2b(const uint8_t *challenge, size_t pow_bitlen, uint8_t *pow)
{
    if (!challenge || !pow) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    // the first batch is small: low difficulties finish in one launch; the later ones are large — a launch stops soon after its first hit
    uint64_t found = ~0ull, first = 0, batch = (uint64_t)1 << 16;
    while (found == ~0ull) {
        const int rc = iopx_pow_search_blake2b(challenge, pow_bitlen, first, batch, &found);
        if (rc != IOPX_OK) return rc;
        first += batch;
        if (batch < ((uint64_t)1 << 28)) batch <<= 4;
    }
    return iopx_pow_candidate_blake2b(challenge, found, pow);
}

// merkle_tree::get_set_membership_proof (libiop/bcs/merkle_tree.tcc:242-336) on a device-resident node array.
int iopx_merkle_membership_proof_dev(const uint8_t *d_nodes, size_t num_leaves, const size_t *positions, size_t num_positions,
                                     uint8_t *aux_hashes, size_t aux_capacity, size_t *num_aux)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_nodes || !num_aux || (num_positions && !positions)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (num_leaves < 2 || (num_leaves & (num_leaves - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "Merkle tree size must be a power of two, and at least 2.");
    *num_aux = 0;
    if (num_positions == 0) return IOPX_OK;
    // the sorted set of queried nodes of one level; a node whose sibling is not in the set contributes the sibling
    std::vector<uint64_t> level(positions, positions + num_positions), want;
    std::sort(level.begin(), level.end());
    level.erase(std::unique(level.begin(), level.end()), level.end());
    if (level.back() >= num_leaves) return fail(IOPX_ERR_INVALID_ARGUMENT, "All positions must be between 0 and num_leaves-1.");
    for (uint64_t &v : level) v += num_leaves - 1;
    while (!(level.size() == 1 && level[0] == 0)) {
        std::vector<uint64_t> parents;
        for (size_t i = 0; i < level.size(); ++i) {
            const uint64_t node = level[i];
            parents.push_back((node - 1) / 2);
            if ((node & 1) == 0) want.push_back(node - 1);
            else if (i + 1 < level.size() && level[i + 1] == node + 1) ++i;
            else want.push_back(node + 1);
        }
        level.swap(parents);
    }
    *num_aux = want.size();
    if (want.empty()) return IOPX_OK;
    if (!aux_hashes || aux_capacity < want.size()) return fail(IOPX_ERR_INVALID_ARGUMENT, "auxiliary hash buffer too small: %zu needed", want.size());
    TmpBuf didx, dout;
    if ((rc = didx.alloc(want.size() * 8)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(want.size() * 32)) != IOPX_OK) return rc;
    if ((rc = upload(didx.p, want.data(), want.size() * 8)) != IOPX_OK) return rc;
    { ProfScope ps_("k_gather_nodes");
      hipLaunchKernelGGL(k_gather_nodes, dim3((unsigned)((4 * want.size() + 255) / 256)), dim3(256), 0, stream(), dout.u64(), (const uint64_t *)d_nodes,
                         (const uint64_t *)didx.u64(), want.size()); }
    { int drc_ = download(aux_hashes, dout.p, want.size() * 32, /*deferrable=*/true); if (drc_ != IOPX_OK) return drc_; }
    return IOPX_OK;
}

// The query responses of bcs_prover::get_transcript (libiop/bcs/bcs_prover.tcc:187-197): values[p][k] = oracle_k[positions[p]].
int iopx_query_responses_dev(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n, const size_t *positions,
                             size_t num_positions, void *values)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_oracles || num_oracles == 0 || (num_positions && (!positions || !values))) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (elem_bytes == 0 || (elem_bytes & 7)) return fail(IOPX_ERR_INVALID_ARGUMENT, "element size %zu is not a multiple of 8 bytes", elem_bytes);
    if (num_positions == 0) return IOPX_OK;
    std::vector<uint64_t> pos(positions, positions + num_positions);
    for (uint64_t v : pos) if (v >= n) return fail(IOPX_ERR_INVALID_ARGUMENT, "query position %llu outside the domain", (unsigned long long)v);
    const size_t words = elem_bytes / 8, total = num_positions * num_oracles * words;
    TmpBuf dptrs, dpos, dout;
    if ((rc = dout.alloc(total * 8)) != IOPX_OK) return rc;
    if (num_oracles <= GATHER_INLINE_ORACLES && num_positions <= GATHER_INLINE_POSITIONS) {
        GatherInline g;
        memset(&g, 0, sizeof(g));
        for (size_t k = 0; k < num_oracles; ++k) g.oracles[k] = (const uint64_t *)d_oracles[k];
        memcpy(g.pos, pos.data(), num_positions * 8);
        { ProfScope ps_("k_gather_responses");
          hipLaunchKernelGGL(k_gather_responses_inline, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, stream(), dout.u64(), g,
                             num_oracles, words, num_positions); }
        IOPX_HIP(hipGetLastError());
        { int drc_ = download(values, dout.p, total * 8, /*deferrable=*/true); if (drc_ != IOPX_OK) return drc_; }
        return IOPX_OK;
    }
    if ((rc = dptrs.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = dpos.alloc(num_positions * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dptrs.p, d_oracles, num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = upload(dpos.p, pos.data(), num_positions * 8)) != IOPX_OK) return rc;
    { ProfScope ps_("k_gather_responses");
      hipLaunchKernelGGL(k_gather_responses, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, stream(), dout.u64(),
                         (const uint64_t *const *)dptrs.p, (const uint64_t *)dpos.u64(), num_oracles, words, num_positions); }
    { int drc_ = download(values, dout.p, total * 8, /*deferrable=*/true); if (drc_ != IOPX_OK) return drc_; }
    return IOPX_OK;
}

} // extern "C"
