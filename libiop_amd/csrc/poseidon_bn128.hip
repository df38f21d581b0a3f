// Poseidon leaf / two-to-one hashing and the BCS Merkle tree built on it, over alt_bn128 Fr, for gfx950.
//
// Replaces, for hash_enum = starkware_poseidon_type / high_alpha_poseidon_type (the only field the reference wires
// Poseidon for, libiop/bcs/hashing/hash_enum.tcc:12-24,73-165):
//   poseidon::apply_permutation          libiop/bcs/hashing/poseidon.tcc:159-297 (alpha in {3,5,17}; dense MDS or the add-only
//                                        near-MDS forms of :195-239)
//   algebraic_sponge::absorb / squeeze   libiop/bcs/hashing/algebraic_sponge.tcc:18-100
//   algebraic_leafhash::hash / zk_hash   algebraic_sponge.tcc:220-245 (salt parsing :110-125)
//   algebraic_two_to_one_hash::hash      algebraic_sponge.tcc:256-265
//   merkle_tree::construct_with_leaves_serialized_by_cosets + compute_inner_nodes (merkle_tree.tcc:92-151,200-229)
//
// Elements and digests are libff Fp_model Montgomery words (4 x uint64, R = 2^256).  One lane hashes one leaf / node; the
// arithmetic runs on nine 29-bit limbs (see "alt_bn128 Fr in radix 2^29" below).  The parameter set (round constants, MDS)
// is supplied by the caller as canonical integers — libiop holds them in poseidon_params (poseidon.tcc:311-520) — converted
// to the internal form on the device and cached.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "runtime.h"
#include "poseidon_tables.h"

namespace iopx {

// ---- alt_bn128 Fr in radix 2^29 ----------------------------------------------------------------------------------
// gfx950's widest integer multiply is v_mad_u64_u32 (32 x 32 + 64 -> 64, no carry-in).  Nine 29-bit limbs leave enough
// headroom in that 64-bit accumulator for a whole column of a Montgomery product (nine a_i * b_j and nine m_i * p_j
// terms), so a product is 162 multiply-adds and no carry chains; additions are limb-wise with no carries at all.
// Internally an element x is any representative of x * 2^261 mod p below 2^257.5 ("R261 form"); the library's
// boundary stays libff's 4 x 64-bit Montgomery words (x * 2^256 mod p, canonical), converted on load / store.
struct bn9 {
    uint32_t l[9];
};

#define BN9_MASK 0x1fffffffu
__device__ static const uint32_t BN9_P[9] = { 0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu };
#define BN9_INV 0x0fffffffu                  // -p^-1 mod 2^29
// 2^e mod p as plain 29-bit limbs: multiplying by them (one Montgomery product, / 2^261) moves between the forms
__device__ static const uint32_t BN9_C256[9] = { 0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u };
__device__ static const uint32_t BN9_C266[9] = { 0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u };
__device__ static const uint32_t BN9_C517[9] = { 0x142db4dfu, 0x19d6990eu, 0x1472f48cu, 0x06dbe7e3u, 0x0b84d579u, 0x10f9faf7u, 0x121f4380u, 0x17a112deu, 0x001275c7u };
__device__ static const uint32_t BN9_C522[9] = { 0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au };
__device__ static const uint32_t BN_P32[8] = { 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u };

__device__ __forceinline__ bn9 bn9_const(const uint32_t (&c)[9])
{
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c[i];
    return r;
}

__device__ __forceinline__ bn9 bn9_zero()
{
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}

// 256-bit integer (4 little-endian 64-bit words) -> nine 29-bit limbs
__device__ __forceinline__ bn9 bn9_unpack(const uint64_t *q)
{
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint64_t v = q[i]; w[2 * i] = (uint32_t)v; w[2 * i + 1] = (uint32_t)(v >> 32); }
    w[8] = 0;
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)w[wi] | ((uint64_t)w[wi + 1 > 8 ? 8 : wi + 1] << 32);
        r.l[i] = (uint32_t)(two >> sh) & BN9_MASK;
    }
    return r;
}

// limb-wise sum; no carries (see the headroom rules at bn9_mul)
__device__ __forceinline__ bn9 bn9_add(const bn9 &a, const bn9 &b)
{
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// carry propagation: same value, limbs 0..7 back below 2^29
__device__ __forceinline__ bn9 bn9_norm(const bn9 &a)
{
    bn9 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t v = a.l[i] + c;
        r.l[i] = v & BN9_MASK;
        c = v >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// 2^254 - p, plain limbs: v = q * 2^254 + rem is congruent to q * (2^254 - p) + rem
__device__ static const uint32_t BN9_C254[9] = { 0x0fffffffu, 0x00f05360u, 0x11a3dbafu, 0x182f6f0cu, 0x0a7a2d7cu, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x000f9bb1u };

// Weak reduction of any limb vector (limbs up to 2^32 - 1): same residue, value below 2^254 + (v >> 254) * 0.25 * 2^254 and
// normalised limbs.  Used where elements are only ever added (the near-MDS layers), so that they stay below 2^256.
__device__ __forceinline__ bn9 bn9_reduce(const bn9 &a)
{
    bn9 n = bn9_norm(a);
    const uint32_t q = n.l[8] >> 22;
    n.l[8] &= 0x3fffffu;
    bn9 r;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc += (uint64_t)q * BN9_C254[i] + n.l[i];
        r.l[i] = (uint32_t)acc & BN9_MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc + q * BN9_C254[8] + n.l[8];
    return r;
}

// Montgomery reduction interleaved with the column sums (product scanning).  On entry to column k `acc` holds the
// carry of column k - 1; COLUMN(k) adds the operand products, then the m_i * p_j terms.
#define BN9_REDUCE_LOW(k)                                                                 \
    {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < (k); ++i) acc += (uint64_t)m[i] * BN9_P[(k) - i]; \
        m[k] = ((uint32_t)acc * BN9_INV) & BN9_MASK;                                      \
        acc += (uint64_t)m[k] * BN9_P[0];                                                 \
        acc >>= 29;                                                                       \
    }
#define BN9_REDUCE_HIGH(k)                                                                \
    {                                                                                     \
        _Pragma("unroll") for (int i = (k) - 8; i < 9; ++i) acc += (uint64_t)m[i] * BN9_P[(k) - i]; \
        r.l[(k) - 9] = (uint32_t)acc & BN9_MASK;                                          \
        acc >>= 29;                                                                       \
    }

// sum_n a[n] * b[n] * 2^-261 mod p for N operand pairs reduced together.
// Headroom: every column must stay below 2^64, i.e. N * 9 * max(a limb) * max(b limb) + 9 * 2^58 < 2^64:
//   N = 1: both operands may have limbs up to 2^30 (one carry-less addition each), or one up to 2^31 against a normalised one;
//   N = 3: one side normalised (< 2^29), the other up to 2^30;   N = 4: both normalised.
// Values: inputs below 2^257.5 give a result below 2^255 with normalised limbs.
template<int N>
__device__ __forceinline__ bn9 bn9_dot(const bn9 (&a)[N], const bn9 (&b)[N])
{
    uint32_t m[9];
    bn9 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a[n].l[i] * b[n].l[k - i];
        }
        BN9_REDUCE_LOW(k)
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a[n].l[i] * b[n].l[k - i];
        }
        BN9_REDUCE_HIGH(k)
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

__device__ __forceinline__ bn9 bn9_mul(const bn9 &a, const bn9 &b)
{
    const bn9 aa[1] = { a }, bb[1] = { b };
    return bn9_dot<1>(aa, bb);
}

// a * a * 2^-261: the cross terms once, against the doubled operand (45 products instead of 81); limbs of a up to 2^30
__device__ __forceinline__ bn9 bn9_sqr(const bn9 &a)
{
    uint32_t m[9], d[9];
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
        BN9_REDUCE_LOW(k)
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
        BN9_REDUCE_HIGH(k)
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// libff words (x * 2^256, canonical) -> R261 form
__device__ __forceinline__ bn9 bn9_load_mont(const uint64_t *p, size_t idx)
{
    return bn9_mul(bn9_unpack(p + 4 * idx), bn9_const(BN9_C266));
}

// value below 2^255 with normalised limbs -> canonical 4 x 64-bit words
__device__ __forceinline__ void bn9_store_canonical(uint64_t *q, const bn9 &y)
{
    uint32_t w[8], d[8];
    uint64_t acc = 0;
    int bits = 0, wi = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        acc |= (uint64_t)y.l[i] << bits;
        bits += 29;
        if (bits >= 32 && wi < 8) { w[wi++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
    }
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t t = (uint64_t)w[i] - BN_P32[i] - borrow;
        d[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = borrow ? w[2 * i] : d[2 * i], hi = borrow ? w[2 * i + 1] : d[2 * i + 1];
        q[i] = (uint64_t)lo | ((uint64_t)hi << 32);
    }
}

// R261 form (any representative) -> libff words
__device__ __forceinline__ void bn9_store_mont(uint64_t *p, size_t idx, const bn9 &x)
{
    bn9_store_canonical(p + 4 * idx, bn9_mul(x, bn9_const(BN9_C256)));
}

// device copy of a parameter set: ark[(R_F + R_P) * t] then mds[t * t], nine 29-bit limbs each, R261 form
struct PoseidonDev {
    const uint32_t *consts;
    int alpha, full_rounds, partial_rounds, rate, t, near_mds;
};

__device__ __forceinline__ bn9 bn9_load_const(const uint32_t *c, size_t idx)
{
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c[9 * idx + i];
    return r;
}

// x^alpha (poseidon.tcc:159-193); x may carry one pending addition (limbs < 2^30)
__device__ __forceinline__ bn9 poseidon_sbox(const bn9 &x, int alpha)
{
    bn9 t = bn9_sqr(x);
    if (alpha >= 5) t = bn9_sqr(t);
    if (alpha == 17) { t = bn9_sqr(t); t = bn9_sqr(t); }
    return bn9_mul(t, x);
}

template<int T>
struct pstate {
    bn9 e[T];       // only ever indexed by constants: stays in registers
};

template<int T>
__device__ __forceinline__ void poseidon_mix(pstate<T> &s, const PoseidonDev &P, bool full)
{
    if (P.near_mds) {
        // add-only layers; the sums stay carry-less and are normalised at the next round's constant addition
        if constexpr (T == 3) {             // poseidon.tcc:198-211
            const bn9 x = s.e[0];
            s.e[0] = bn9_add(s.e[0], s.e[2]);
            s.e[2] = bn9_add(s.e[2], s.e[1]);
            s.e[1] = bn9_add(s.e[1], x);
        } else {                            // :213-224: every element becomes the sum of the other three
            const bn9 a = bn9_add(s.e[0], s.e[1]), b = bn9_add(s.e[2], s.e[3]);
            const bn9 n0 = bn9_add(s.e[1], b), n1 = bn9_add(s.e[0], b), n2 = bn9_add(s.e[3], a), n3 = bn9_add(s.e[2], a);
            s.e[0] = n0; s.e[1] = n1; s.e[2] = n2; s.e[3] = n3;
        }
    } else {                                // :226-238 dense MDS: one reduction per output row
        const uint32_t *mds = P.consts + 9 * (size_t)(P.full_rounds + P.partial_rounds) * T;
        if (!full) {
            // elements that skipped the S-box still carry their round-constant addition
            s.e[0] = bn9_norm(s.e[0]);
            s.e[1] = bn9_norm(s.e[1]);
            if constexpr (T == 4) s.e[2] = bn9_norm(s.e[2]);
        }
        pstate<T> o;
        if constexpr (T == 3) {
            const bn9 v[3] = { s.e[0], s.e[1], s.e[2] };
            { const bn9 r[3] = { bn9_load_const(mds, 0), bn9_load_const(mds, 1), bn9_load_const(mds, 2) }; o.e[0] = bn9_dot<3>(r, v); }
            { const bn9 r[3] = { bn9_load_const(mds, 3), bn9_load_const(mds, 4), bn9_load_const(mds, 5) }; o.e[1] = bn9_dot<3>(r, v); }
            { const bn9 r[3] = { bn9_load_const(mds, 6), bn9_load_const(mds, 7), bn9_load_const(mds, 8) }; o.e[2] = bn9_dot<3>(r, v); }
        } else {
            const bn9 v[4] = { s.e[0], s.e[1], s.e[2], s.e[3] };
            { const bn9 r[4] = { bn9_load_const(mds, 0), bn9_load_const(mds, 1), bn9_load_const(mds, 2), bn9_load_const(mds, 3) }; o.e[0] = bn9_dot<4>(r, v); }
            { const bn9 r[4] = { bn9_load_const(mds, 4), bn9_load_const(mds, 5), bn9_load_const(mds, 6), bn9_load_const(mds, 7) }; o.e[1] = bn9_dot<4>(r, v); }
            { const bn9 r[4] = { bn9_load_const(mds, 8), bn9_load_const(mds, 9), bn9_load_const(mds, 10), bn9_load_const(mds, 11) }; o.e[2] = bn9_dot<4>(r, v); }
            { const bn9 r[4] = { bn9_load_const(mds, 12), bn9_load_const(mds, 13), bn9_load_const(mds, 14), bn9_load_const(mds, 15) }; o.e[3] = bn9_dot<4>(r, v); }
        }
        s = o;
    }
}

// poseidon::apply_permutation (poseidon.tcc:273-297).  On entry: the previous mixing layer's output (a normalised value,
// or a carry-less sum of up to three values below 2^256), possibly plus one absorbed element; on exit: the same.
template<int T>
__device__ __forceinline__ void poseidon_permute(pstate<T> &s, const PoseidonDev &P)
{
    const int half = P.full_rounds / 2, total = P.full_rounds + P.partial_rounds;
#pragma unroll 1
    for (int round = 0; round < total; ++round) {
        const bool full = round < half || round >= half + P.partial_rounds;
        const uint32_t *rc = P.consts + 9 * (size_t)round * T;
        s.e[0] = bn9_add(s.e[0], bn9_load_const(rc, 0));
        s.e[1] = bn9_add(s.e[1], bn9_load_const(rc, 1));
        s.e[2] = bn9_add(s.e[2], bn9_load_const(rc, 2));
        if constexpr (T == 4) s.e[3] = bn9_add(s.e[3], bn9_load_const(rc, 3));
        if (P.near_mds || round == 0) {
            // sums of up to three elements plus the constant (round 0: plus what the caller absorbed).  Elements that skip
            // the S-box are never multiplied in the add-only layers, so this is also where they are kept below 2^256.
            s.e[0] = bn9_reduce(s.e[0]);
            s.e[1] = bn9_reduce(s.e[1]);
            s.e[2] = bn9_reduce(s.e[2]);
            if constexpr (T == 4) s.e[3] = bn9_reduce(s.e[3]);
        }
        if (full) {
            s.e[0] = poseidon_sbox(s.e[0], P.alpha);
            s.e[1] = poseidon_sbox(s.e[1], P.alpha);
            if constexpr (T == 4) s.e[2] = poseidon_sbox(s.e[2], P.alpha);
        }
        s.e[T - 1] = poseidon_sbox(s.e[T - 1], P.alpha);
        poseidon_mix<T>(s, P, full);
    }
}

// 4-word integers -> R261 form (consts for the kernels above); also reduces values >= p like FieldT(bigint)
__global__ void k_bn_to_r261(uint32_t *out, const uint64_t *in, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const bn9 v = bn9_mul(bn9_unpack(in + 4 * i), bn9_const(BN9_C522));
#pragma unroll
        for (int k = 0; k < 9; ++k) out[9 * i + k] = v.l[k];
    }
}

// canonical 4-word integers -> libff Montgomery words
__global__ void k_bn_to_mont(uint64_t *out, const uint64_t *in, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        bn9_store_canonical(out + 4 * i, bn9_mul(bn9_unpack(in + 4 * i), bn9_const(BN9_C517)));
    }
}

template<int T>
__global__ void k_poseidon_permute(PoseidonDev P, uint64_t *states, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        pstate<T> s;
        s.e[0] = bn9_load_mont(states, i * T);
        s.e[1] = bn9_load_mont(states, i * T + 1);
        s.e[2] = bn9_load_mont(states, i * T + 2);
        if constexpr (T == 4) s.e[3] = bn9_load_mont(states, i * T + 3);
        poseidon_permute<T>(s, P);
        bn9_store_mont(states, i * T, s.e[0]);
        bn9_store_mont(states, i * T + 1, s.e[1]);
        bn9_store_mont(states, i * T + 2, s.e[2]);
        if constexpr (T == 4) bn9_store_mont(states, i * T + 3, s.e[3]);
    }
}

struct PLeafParams {
    const uint64_t *const *oracles;
    const uint64_t *salts;          // nullptr, or num_leaves * 32 bytes
    uint64_t *nodes;                // (2 L - 1) elements
    size_t num_oracles, n, coset_size, num_leaves;
    int additive;
};

// element g of leaf i's serialisation: slice[j + k * coset_size] = oracle_k[pos(i, j)] (merkle_tree.tcc:127-134), then
// the zk salt parsed as string_to_field_elem does (8-byte words, first word most significant, then FieldT(bigint))
__device__ __forceinline__ bn9 poseidon_leaf_element(const PLeafParams &p, size_t i, size_t g)
{
    if (g < p.num_oracles * p.coset_size) {
        const size_t ko = g / p.coset_size, j = g % p.coset_size;
        const size_t pos = p.additive ? i * p.coset_size + j : i + j * p.num_leaves;
        return bn9_load_mont(p.oracles[ko], pos);
    }
    const uint64_t *sp = p.salts + 4 * i;
    const uint64_t rev[4] = { sp[3], sp[2], sp[1], sp[0] };
    return bn9_mul(bn9_unpack(rev), bn9_const(BN9_C522));
}

// algebraic_leafhash::hash / zk_hash: absorb the slice rate-wise (algebraic_sponge.tcc:32-62), squeeze element 0
template<int T>
__global__ void k_poseidon_leaves(PoseidonDev P, PLeafParams p)
{
    constexpr int RATE = T - 1;
    const size_t len = p.num_oracles * p.coset_size + (p.salts ? 1 : 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.num_leaves; i += (size_t)gridDim.x * blockDim.x) {
        pstate<T> s;
        s.e[0] = bn9_zero(); s.e[1] = bn9_zero(); s.e[2] = bn9_zero();
        if constexpr (T == 4) s.e[3] = bn9_zero();
        // every chunk of `rate` elements is followed by exactly one permutation: the absorb loop's for all but the last
        // chunk, the squeeze's (algebraic_sponge.tcc:78-90) for the last
#pragma unroll 1
        for (size_t begin = 0; ; begin += RATE) {
            const size_t left = len - begin;
            s.e[0] = bn9_add(s.e[0], poseidon_leaf_element(p, i, begin));
            if (left > 1) s.e[1] = bn9_add(s.e[1], poseidon_leaf_element(p, i, begin + 1));
            if constexpr (T == 4) { if (left > 2) s.e[2] = bn9_add(s.e[2], poseidon_leaf_element(p, i, begin + 2)); }
            poseidon_permute<T>(s, P);
            if (left <= (size_t)RATE) break;
        }
        bn9_store_mont(p.nodes, p.num_leaves - 1 + i, s.e[0]);
    }
}

template<int T>
__device__ __forceinline__ void poseidon_node(const PoseidonDev &P, uint64_t *nodes, size_t j)
{
    pstate<T> s;
    s.e[0] = bn9_load_mont(nodes, 2 * j + 1);   // algebraic_two_to_one_hash::hash (algebraic_sponge.tcc:256-265)
    s.e[1] = bn9_load_mont(nodes, 2 * j + 2);
    s.e[2] = bn9_zero();
    if constexpr (T == 4) s.e[3] = bn9_zero();
    poseidon_permute<T>(s, P);
    bn9_store_mont(nodes, j, s.e[0]);
}

template<int T>
__global__ void k_poseidon_level(PoseidonDev P, uint64_t *nodes, size_t first, size_t count)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x) {
        poseidon_node<T>(P, nodes, first + j);
    }
}

template<int T>
__global__ void k_poseidon_top(PoseidonDev P, uint64_t *nodes, size_t count)
{
    for (size_t c = count; c >= 1; c >>= 1) {
        for (size_t j = threadIdx.x; j < c; j += blockDim.x) poseidon_node<T>(P, nodes, (c - 1) + j);
        __threadfence_block();
        __syncthreads();
    }
}

// ---- small tree levels: one permutation spread over t lanes ------------------------------------------------------------
// A level with few nodes runs at the LATENCY of one permutation (about 600 dependent field products on one lane).  Here lane
// (node, e) owns state element e: the S-boxes of a full round and the t output rows of the mixing layer run side by side, the
// elements cross lanes through LDS (two barriers per round).  About twice faster per level for the dense-MDS sets; partial
// rounds still wait for the one S-box, so large levels keep the one-lane-per-node kernel (better throughput).
#define POSEIDON_PAR_NODES 64           // nodes per workgroup (64 * t lanes = t waves: they spread over the CU's four SIMDs)

__device__ __forceinline__ uint32_t *par_slot(uint32_t *lds, int T, int buf, int e, int node) { return lds + ((size_t)(buf * T + e) * 9) * POSEIDON_PAR_NODES + node; }
__device__ __forceinline__ bn9 par_get(uint32_t *lds, int T, int buf, int e, int node)
{
    const uint32_t *q = par_slot(lds, T, buf, e, node);
    bn9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = q[i * POSEIDON_PAR_NODES];
    return r;
}
__device__ __forceinline__ void par_put(uint32_t *lds, int T, int buf, int e, int node, const bn9 &v)
{
    uint32_t *q = par_slot(lds, T, buf, e, node);
#pragma unroll
    for (int i = 0; i < 9; ++i) q[i * POSEIDON_PAR_NODES] = v.l[i];
}

// two-to-one hashes of nodes first .. first + nn - 1 (nn <= POSEIDON_PAR_NODES) by the whole workgroup
template<int T>
__device__ __forceinline__ void poseidon_nodes_par(const PoseidonDev &P, uint64_t *nodes, size_t first, int nn, uint32_t *lds)
{
    const int lanes = nn * T;
    const int nn_log = 31 - __builtin_clz((unsigned)nn);        // nn is a power of two (tree levels are)
    for (int t = threadIdx.x; t < lanes; t += blockDim.x) {
        const int node = t & (nn - 1), e = t >> nn_log;
        // state[0] = left child, state[1] = right child (algebraic_sponge.tcc:256-265), the rest zero
        par_put(lds, T, 0, e, node, e < 2 ? bn9_load_mont(nodes, 2 * (first + node) + 1 + e) : bn9_zero());
    }
    __syncthreads();
    const int half = P.full_rounds / 2, total = P.full_rounds + P.partial_rounds;
    const uint32_t *mds = P.consts + 9 * (size_t)total * T;
    for (int round = 0; round < total; ++round) {
        const bool full = round < half || round >= half + P.partial_rounds;
        for (int t = threadIdx.x; t < lanes; t += blockDim.x) {                 // constants and S-boxes (poseidon.tcc:241-271)
            const int node = t & (nn - 1), e = t >> nn_log;
            bn9 x = bn9_add(par_get(lds, T, 0, e, node), bn9_load_const(P.consts, (size_t)round * T + e));
            if (P.near_mds || round == 0) x = bn9_reduce(x);
            if (full || e == T - 1) x = poseidon_sbox(x, P.alpha);
            else if (!P.near_mds) x = bn9_norm(x);
            par_put(lds, T, 1, e, node, x);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < lanes; t += blockDim.x) {                 // mixing layer, one output row per lane (:195-239)
            const int node = t & (nn - 1), e = t >> nn_log;
            bn9 out;
            if (P.near_mds) {
                if constexpr (T == 3) {
                    // new0 = s0 + s2, new1 = s1 + s0, new2 = s2 + s1
                    out = bn9_add(par_get(lds, T, 1, e, node), par_get(lds, T, 1, (e + 2) % 3, node));
                } else {
                    out = bn9_zero();
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (c != e) out = bn9_add(out, par_get(lds, T, 1, c, node));
                }
            } else {
                bn9 row[T], v[T];
#pragma unroll
                for (int c = 0; c < T; ++c) { row[c] = bn9_load_const(mds, e * T + c); v[c] = par_get(lds, T, 1, c, node); }
                out = bn9_dot<T>(row, v);
            }
            par_put(lds, T, 0, e, node, out);
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < nn; t += blockDim.x) bn9_store_mont(nodes, first + t, par_get(lds, T, 0, 0, t));
}

template<int T>
__global__ void __launch_bounds__(POSEIDON_PAR_NODES * T) k_poseidon_level_par(PoseidonDev P, uint64_t *nodes, size_t first, size_t count)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    for (size_t base = (size_t)blockIdx.x * POSEIDON_PAR_NODES; base < count; base += (size_t)gridDim.x * POSEIDON_PAR_NODES) {
        const int nn = count - base < POSEIDON_PAR_NODES ? (int)(count - base) : POSEIDON_PAR_NODES;
        poseidon_nodes_par<T>(P, nodes, first + base, nn, (uint32_t *)iopx_smem);
        __syncthreads();
    }
}

// the top of the tree in one workgroup: levels of `count`, count / 2, ..., 1 nodes
template<int T>
__global__ void __launch_bounds__(POSEIDON_PAR_NODES * T) k_poseidon_top_par(PoseidonDev P, uint64_t *nodes, size_t count)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    for (size_t c = count; c >= 1; c >>= 1) {
        poseidon_nodes_par<T>(P, nodes, c - 1, (int)c, (uint32_t *)iopx_smem);
        __threadfence_block();
        __syncthreads();
    }
}

// Proof-of-work grind, algebraic digests (libiop/bcs/pow.tcc:73-84,129-141): candidate k is FieldT(k); it passes when word 0 of
// the canonical integer of two_to_one(challenge, k) has its low `bitlen` bits zero.  One lane per candidate, smallest index wins.
struct PowChallengeBn {
    uint64_t w[4];
};

template<int T>
__global__ void k_pow_poseidon(PoseidonDev P, PowChallengeBn c, uint64_t first, uint64_t count, uint64_t mask, unsigned long long *best)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t idx = first + g;
        const uint64_t k[4] = { idx, 0, 0, 0 };
        pstate<T> s;
        s.e[0] = bn9_load_mont(c.w, 0);
        s.e[1] = bn9_mul(bn9_unpack(k), bn9_const(BN9_C522));
        s.e[2] = bn9_zero();
        if constexpr (T == 4) s.e[3] = bn9_zero();
        poseidon_permute<T>(s, P);
        bn9 one = bn9_zero();
        one.l[0] = 1;
        uint64_t canon[4];
        bn9_store_canonical(canon, bn9_mul(s.e[0], one));       // x * 2^261 / 2^261
        if ((canon[0] & mask) == 0) atomicMin(best, (unsigned long long)idx);
    }
}

// ---- host: parameter cache ---------------------------------------------------------------------------------------
struct PoseidonSet {
    DevBuf consts;
    PoseidonDev dev;
};
static std::mutex g_pos_mu;
static std::map<std::vector<uint64_t>, std::unique_ptr<PoseidonSet>> g_pos_sets;

void clear_poseidon_sets()
{
    std::lock_guard<std::mutex> lk(g_pos_mu);
    g_pos_sets.clear();
}

static int pgrid(size_t work, int threads)
{
    size_t g = (work + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > 65536) g = 65536;
    return (int)g;
}

static int get_poseidon(const iopx_poseidon_params *pp, PoseidonDev *out)
{
    if (!pp || !pp->ark || (!pp->mds && !pp->near_mds)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null Poseidon parameters");
    const size_t t = pp->state_size, rounds = pp->full_rounds + pp->partial_rounds;
    if (t != 3 && t != 4) return fail(IOPX_ERR_INVALID_ARGUMENT, "unsupported Poseidon state size %zu (3 or 4)", t);
    if (pp->alpha != 3 && pp->alpha != 5 && pp->alpha != 17) return fail(IOPX_ERR_INVALID_ARGUMENT, "unsupported Poseidon alpha %zu", pp->alpha);
    if (t - pp->rate != 1) return fail(IOPX_ERR_INVALID_ARGUMENT, "Poseidon capacity must be 1 (algebraic_sponge.tcc:212)");
    std::vector<uint64_t> key = { (uint64_t)pp->alpha, (uint64_t)pp->full_rounds, (uint64_t)pp->partial_rounds, (uint64_t)pp->rate, (uint64_t)t, (uint64_t)(pp->near_mds != 0) };
    key.insert(key.end(), pp->ark, pp->ark + 4 * rounds * t);
    if (pp->near_mds) key.insert(key.end(), 4 * t * t, 0);
    else key.insert(key.end(), pp->mds, pp->mds + 4 * t * t);
    std::lock_guard<std::mutex> lk(g_pos_mu);
    auto it = g_pos_sets.find(key);
    if (it == g_pos_sets.end()) {
        std::unique_ptr<PoseidonSet> ps(new PoseidonSet());
        const size_t count = rounds * t + t * t;
        int rc = ps->consts.alloc(count * 36);
        if (rc != IOPX_OK) return rc;
        TmpBuf raw;
        if ((rc = raw.alloc(count * 32)) != IOPX_OK) return rc;
        if ((rc = upload(raw.p, key.data() + 6, count * 32)) != IOPX_OK) return rc;
        { ProfScope ps_("k_bn_to_r261"); hipLaunchKernelGGL(k_bn_to_r261, dim3(pgrid(count, 64)), dim3(64), 0, stream(), (uint32_t *)ps->consts.p, (const uint64_t *)raw.u64(), count); }
        IOPX_HIP(hipStreamSynchronize(stream()));
        ps->dev.consts = (const uint32_t *)ps->consts.p;
        ps->dev.alpha = (int)pp->alpha; ps->dev.full_rounds = (int)pp->full_rounds; ps->dev.partial_rounds = (int)pp->partial_rounds;
        ps->dev.rate = (int)pp->rate; ps->dev.t = (int)t; ps->dev.near_mds = pp->near_mds != 0;
        it = g_pos_sets.emplace(key, std::move(ps)).first;
    }
    *out = it->second->dev;
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

// get_poseidon_parameters (hash_enum.tcc:12-24): 2 = starkware_poseidon_type (alpha 5, t = 3), 3 = high_alpha_poseidon_type
// (alpha 17, state_size 3 by default, 4 on request: poseidon.hpp:56).  Host-only: the tables are static data of the library.
int iopx_poseidon_shipped_params(int bcs_hash_type, size_t state_size, iopx_poseidon_params *out)
{
    if (!out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const char *want = nullptr;
    if (bcs_hash_type == 2 && (state_size == 0 || state_size == 3)) want = "starkware_alpha5_t3";
    else if (bcs_hash_type == 3 && (state_size == 0 || state_size == 3)) want = "high_alpha17_t3";
    else if (bcs_hash_type == 3 && state_size == 4) want = "high_alpha17_t4";
    else if (bcs_hash_type == 3) return fail(IOPX_ERR_INVALID_ARGUMENT, "high_alpha_128_bit_altbn_poseidon_params only supports state size 3 or 4");
    else return fail(IOPX_ERR_INVALID_ARGUMENT, "Not a poseidon hash type");
    for (const ShippedPoseidon &s : SHIPPED_POSEIDON) {
        if (strcmp(s.name, want) != 0) continue;
        out->alpha = s.alpha; out->full_rounds = s.full_rounds; out->partial_rounds = s.partial_rounds;
        out->rate = s.rate; out->state_size = s.state_size; out->near_mds = s.near_mds; out->ark = s.ark; out->mds = s.mds;
        return IOPX_OK;
    }
    return fail(IOPX_ERR_LOGIC, "parameter set %s missing from the library", want);
}

int iopx_bn128_to_montgomery_dev(const uint64_t *d_canonical, uint64_t *d_out, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    { ProfScope ps_("k_bn_to_mont"); hipLaunchKernelGGL(k_bn_to_mont, dim3(pgrid(count, 256)), dim3(256), 0, stream(), d_out, d_canonical, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_poseidon_permute_bn128_dev(const iopx_poseidon_params *params, uint64_t *d_states, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    PoseidonDev P;
    if ((rc = get_poseidon(params, &P)) != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    { ProfScope ps_("k_poseidon_permute"); if (P.t == 3) hipLaunchKernelGGL(k_poseidon_permute<3>, dim3(pgrid(count, 64)), dim3(64), 0, stream(), P, d_states, count);
      else hipLaunchKernelGGL(k_poseidon_permute<4>, dim3(pgrid(count, 64)), dim3(64), 0, stream(), P, d_states, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_merkle_poseidon_bn128_dev(const iopx_poseidon_params *params, const void *const *d_oracles, size_t num_oracles, size_t n,
                                   size_t coset_size, int domain_type, const uint8_t *d_salts, uint64_t *d_nodes)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_oracles || num_oracles == 0 || !d_nodes) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (domain_type != IOPX_DOMAIN_ADDITIVE && domain_type != IOPX_DOMAIN_MULTIPLICATIVE)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "unsupported domain type %d", domain_type);
    if (coset_size == 0 || n % coset_size) return fail(IOPX_ERR_LOGIC, "Attempting to construct a Merkle tree with a constituent vector of wrong size");
    const size_t L = n / coset_size;
    if (L < 2 || (L & (L - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "Merkle tree size must be a power of two, and at least 2.");
    PoseidonDev P;
    if ((rc = get_poseidon(params, &P)) != IOPX_OK) return rc;
    TmpBuf dptrs;
    if ((rc = dptrs.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = upload(dptrs.p, d_oracles, num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    PLeafParams p;
    p.oracles = (const uint64_t *const *)dptrs.p;
    p.salts = (const uint64_t *)d_salts;
    p.nodes = d_nodes;
    p.num_oracles = num_oracles; p.n = n; p.coset_size = coset_size; p.num_leaves = L;
    p.additive = (domain_type == IOPX_DOMAIN_ADDITIVE);
    { ProfScope ps_("k_poseidon_leaves"); if (P.t == 3) hipLaunchKernelGGL(k_poseidon_leaves<3>, dim3(pgrid(L, 64)), dim3(64), 0, stream(), P, p);
      else hipLaunchKernelGGL(k_poseidon_leaves<4>, dim3(pgrid(L, 64)), dim3(64), 0, stream(), P, p); }
    // large levels: one lane per node (throughput); levels of at most 2^14 nodes: one permutation over t lanes (latency)
    static const int par_from = 1 << 14;
    const size_t par_lds = (size_t)2 * P.t * 9 * POSEIDON_PAR_NODES * 4;
    if (P.t == 3) { IOPX_HIP(hipFuncSetAttribute((const void *)k_poseidon_level_par<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)par_lds));
                    IOPX_HIP(hipFuncSetAttribute((const void *)k_poseidon_top_par<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)par_lds)); }
    else { IOPX_HIP(hipFuncSetAttribute((const void *)k_poseidon_level_par<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)par_lds));
           IOPX_HIP(hipFuncSetAttribute((const void *)k_poseidon_top_par<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)par_lds)); }
    size_t count = L / 2;
    while (count > POSEIDON_PAR_NODES) {
        if (count > (size_t)par_from) {
            ProfScope ps_("k_poseidon_level");
            if (P.t == 3) hipLaunchKernelGGL(k_poseidon_level<3>, dim3(pgrid(count, 64)), dim3(64), 0, stream(), P, d_nodes, count - 1, count);
            else hipLaunchKernelGGL(k_poseidon_level<4>, dim3(pgrid(count, 64)), dim3(64), 0, stream(), P, d_nodes, count - 1, count);
        } else {
            ProfScope ps_("k_poseidon_level_par");
            const unsigned grid = (unsigned)((count + POSEIDON_PAR_NODES - 1) / POSEIDON_PAR_NODES);
            if (P.t == 3) hipLaunchKernelGGL(k_poseidon_level_par<3>, dim3(grid), dim3(POSEIDON_PAR_NODES * 3), par_lds, stream(), P, d_nodes, count - 1, count);
            else hipLaunchKernelGGL(k_poseidon_level_par<4>, dim3(grid), dim3(POSEIDON_PAR_NODES * 4), par_lds, stream(), P, d_nodes, count - 1, count);
        }
        count >>= 1;
    }
    { ProfScope ps_("k_poseidon_top_par");
      if (P.t == 3) hipLaunchKernelGGL(k_poseidon_top_par<3>, dim3(1), dim3(POSEIDON_PAR_NODES * 3), par_lds, stream(), P, d_nodes, count);
      else hipLaunchKernelGGL(k_poseidon_top_par<4>, dim3(1), dim3(POSEIDON_PAR_NODES * 4), par_lds, stream(), P, d_nodes, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_merkle_poseidon_bn128(const iopx_poseidon_params *params, const void *const *oracles, size_t num_oracles, size_t n,
                               size_t coset_size, int domain_type, const uint8_t *salts, uint64_t *nodes)
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
        if ((rc = bufs.back()->alloc(n * 32)) != IOPX_OK) return rc;
        IOPX_HIP(copy_h2d(bufs.back()->p, oracles[k], n * 32, stream()));
        dptrs.push_back(bufs.back()->p);
    }
    DevBuf dsalt, dnodes;
    if (salts) {
        if ((rc = dsalt.alloc(L * 32)) != IOPX_OK) return rc;
        IOPX_HIP(copy_h2d(dsalt.p, salts, L * 32, stream()));
    }
    if ((rc = dnodes.alloc((2 * L - 1) * 32)) != IOPX_OK) return rc;
    rc = iopx_merkle_poseidon_bn128_dev(params, dptrs.data(), num_oracles, n, coset_size, domain_type,
                                        salts ? (const uint8_t *)dsalt.p : nullptr, dnodes.u64());
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(nodes, dnodes.p, (2 * L - 1) * 32, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

int iopx_pow_solve_poseidon_bn128(const iopx_poseidon_params *params, const uint64_t *challenge, size_t pow_bitlen, uint64_t *pow)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!challenge || !pow) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (pow_bitlen > 30) return fail(IOPX_ERR_INVALID_ARGUMENT, "pow_bitlen %zu: the reference's `1 << pow_bitlen` is an int shift", pow_bitlen);
    PoseidonDev P;
    if ((rc = get_poseidon(params, &P)) != IOPX_OK) return rc;
    PowChallengeBn c;
    memcpy(c.w, challenge, 32);
    const uint64_t mask = ((uint64_t)1 << pow_bitlen) - 1;
    TmpBuf best;                                        // [0]: smallest passing index, [1..4]: scratch for its Montgomery words
    if ((rc = best.alloc(40)) != IOPX_OK) return rc;
    const unsigned long long none = ~0ull;
    if ((rc = upload(best.p, &none, 8)) != IOPX_OK) return rc;
    unsigned long long found = none;
    uint64_t first = 0, batch = (uint64_t)1 << 14;
    while (found == none) {
        const unsigned grid = (unsigned)((batch + 63) / 64 > 65536 ? 65536 : (batch + 63) / 64);
        { ProfScope ps_("k_pow_poseidon");
          if (P.t == 3) hipLaunchKernelGGL(k_pow_poseidon<3>, dim3(grid), dim3(64), 0, stream(), P, c, first, batch, mask, (unsigned long long *)best.p);
          else hipLaunchKernelGGL(k_pow_poseidon<4>, dim3(grid), dim3(64), 0, stream(), P, c, first, batch, mask, (unsigned long long *)best.p); }
        { int drc_ = download(&found, best.p, 8); if (drc_ != IOPX_OK) return drc_; }
        first += batch;
        if (batch < ((uint64_t)1 << 22)) batch <<= 2;
    }
    const uint64_t k[4] = { found, 0, 0, 0 };
    if ((rc = upload((uint64_t *)best.p + 1, k, 32)) != IOPX_OK) return rc;
    hipLaunchKernelGGL(k_bn_to_mont, dim3(1), dim3(64), 0, stream(), (uint64_t *)best.p + 1, (const uint64_t *)best.p + 1, (size_t)1);
    { int drc_ = download(pow, (uint64_t *)best.p + 1, 32); if (drc_ != IOPX_OK) return drc_; }
    return IOPX_OK;
}

} // extern "C"
