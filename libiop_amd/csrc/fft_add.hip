// Additive (Gao–Mateer-equivalent) FFT / IFFT over GF(2^192) for gfx950.
//
// Replaces additive_FFT / additive_IFFT of the reference (libiop/algebra/fft.tcc:39-124, 126-204).  The
// outputs are the unique evaluations / coefficients, so the kernel schedule is free to differ:
//
//   phase 1  (coefficients -> Gao–Mateer basis), in place on 2^d elements, d = ceil(log2 n_coeffs):
//            for each level j: twist by beta_j^(idx >> j) (power table in HBM, built once per domain),
//            then the Taylor-expansion XOR network on index bits (k+1,k), k = d-2 .. j (fft.tcc:62-83).
//            The XOR network runs in LDS tiles: a pass loads 2^(c+A) elements (2^c contiguous columns x
//            2^A rows on A consecutive index bits) and performs every operation whose two index bits
//            lie inside the tile; the last pass fuses all remaining levels.
//   phase 2  (the "unwind" butterflies, fft.tcc:102-120) is run WITHOUT the bit reversal of fft.tcc:99,
//            i.e. in the order where the 2^(d-1-l) butterflies of one block share one twiddle and are
//            contiguous in memory; the bit reversal is folded into the addressing of the last pass.
//            For n_coeffs << 2^m (the prover's low-degree extension) phase 1 runs once on 2^d elements
//            and phase 2 runs once per coset of span(basis[0..d)) — 2^(m-d) independent transforms whose
//            twiddles differ only by a per-level additive constant (the recursed shift is GF(2)-linear
//            in the coset shift).
//
// The IFFT runs the exact inverse schedule.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "gf192_dev.h"
#include "gf192_host.h"
#include <algorithm>
#include "runtime.h"

namespace iopx {

// Tile geometry.  Defaults: 2048-element phase-2 tiles (48 KiB of LDS per workgroup), 1024-element phase-1 and edge tiles.  The IOPX_TILE_BITS /
// IOPX_P1_COLS / IOPX_P2_COLS / IOPX_P2_TOP environment variables override them (read once): used for
// tuning, and by the tests to exercise the multi-pass schedules at small transform sizes.
struct Tuning { int tile_bits, p1_tile_bits, edge_tile_bits, p1_cols, p2_cols, p2_top, comb, p2_threads, p1_fin_tile_bits, p1_fin_cols, small_last, scratch_mb, p1_comb, rs_comb_cap_log2, edge_multi; };
static int env_int(const char *name, int dflt, int lo, int hi) { return opt_range(name, dflt, lo, hi); }      // runtime.h: the options table
static const Tuning &tuning()
{
    static const Tuning t = [] {
        Tuning u;
        u.tile_bits = env_int("IOPX_TILE_BITS", 11, 4, 12);   // 2048-element tiles: two comb workgroups per CU overlap their load / compute phases
        u.p1_tile_bits = (u.tile_bits < 10 ? u.tile_bits : 10);   // phase 1 is latency-bound: smaller tiles, more workgroups per CU
        u.p1_cols = env_int("IOPX_P1_COLS", 3, 0, u.p1_tile_bits - 3);  // strided phase-1 tiles: 2^c contiguous columns
        // the last phase-1 pass runs every remaining level inside its tile (multiplier-bound): it may use a larger, narrower tile
        // (2^11 rows since the top levels' twists run on the comb product: five wave-uniform levels instead of four, k_phase1 4.24 -> 4.11 ms per proof)
        u.p1_fin_tile_bits = (u.tile_bits > u.p1_tile_bits ? u.p1_tile_bits + 1 : u.p1_tile_bits);
        u.p1_fin_cols = (0);   // measured: 4.74 -> 4.59 ms at 2^22 with single-element columns
        // phase-2 upper passes: 2^c contiguous columns.  With c = 6 every row bit of a tile sits at local bit >= 6, so all of a pass's
        // butterflies have wave-uniform twiddles (comb product); c = 4 left the two lowest row bits of each pass on the general product
        // (k_bfly_upper 36.2 -> 33.0 ms per Aurora 2^20 proof, with 96 instead of 65 launches)
        u.p2_cols = env_int("IOPX_P2_COLS", u.tile_bits - 2 < 6 ? u.tile_bits - 2 : 6, 0, u.tile_bits - 2);
        // the edge pass holds the levels whose twiddles are not wave-uniform (pair bits < 6): general multiplier,
        // small tiles for occupancy; 2^p2_top natural-order runs
        u.edge_tile_bits = (u.tile_bits < 10 ? u.tile_bits : 10);
        u.p2_top = env_int("IOPX_P2_TOP", 4, 0, u.edge_tile_bits - 2);
        u.comb = env_int("IOPX_COMB", 1, 0, 1);                         // 1: asm comb multiplier where the twiddle is wave-uniform
        u.p2_threads = (u.comb ? 512 : 1024);
        u.small_last = env_int("IOPX_SMALL_LAST", 1, 0, 1);             // 1: one-word twiddle numerators at the last level where the basis allows
        u.scratch_mb = (256);
        u.rs_comb_cap_log2 = env_int("IOPX_RS_COMB_CAP_LOG2", 22, 0, 30);  // per-coset combined shift terms up to 2^this entries, byte tables beyond
        u.p1_comb = env_int("IOPX_P1_COMB", 1, 0, 1);                   // 1: comb product for the phase-1 twists with a wave-uniform multiplier
        // > 0: the single-polynomial edge passes take this many cosets of one tile position per workgroup (k_bfly_edge_multi): the tile's twiddles
        // are read once into LDS, the cosets differ in one shift term per level
        u.edge_multi = env_int("IOPX_EDGE_MULTI", 4, 0, 64);
        return u;
    }();
    return t;
}
#define TILE_BITS (tuning().tile_bits)
#define P1_COLS (tuning().p1_cols)
#define P1_TILE_BITS (tuning().p1_tile_bits)
#define P2_COLS (tuning().p2_cols)
#define P2_TOP (tuning().p2_top)
static const int BLOCK_THREADS = 512;
#define SCRATCH_BYTES ((size_t)tuning().scratch_mb << 20)      // block-order staging for a group of LDE cosets

// ---------------------------------------------------------------------------------------------
// LDS tile: three planes of 64-bit words (conflict-free for consecutive element indices)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ gf192 lds_get(const uint64_t *s, int E, int li)
{
    const uint64_t a = s[li], b = s[E + li], c = s[2 * E + li];
    gf192 r;
    r.w[0] = (uint32_t)a; r.w[1] = (uint32_t)(a >> 32);
    r.w[2] = (uint32_t)b; r.w[3] = (uint32_t)(b >> 32);
    r.w[4] = (uint32_t)c; r.w[5] = (uint32_t)(c >> 32);
    return r;
}

__device__ __forceinline__ void lds_put(uint64_t *s, int E, int li, const gf192 &v)
{
    s[li] = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32);
    s[E + li] = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32);
    s[2 * E + li] = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
}

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits)
{
    return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

// ---------------------------------------------------------------------------------------------
// plan construction kernels
// ---------------------------------------------------------------------------------------------
// out[q] = prod_{k : bit k of q} sq[k]  for q < count   (sq[k] = beta^(2^k))
__global__ void k_pow_direct(uint64_t *out, const uint64_t *sq, int nbits, size_t count)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (size_t)gridDim.x * blockDim.x) {
        gf192 acc = gf_zero();
        acc.w[0] = 1;
        for (int k = 0; k < nbits; ++k) {
            if ((q >> k) & 1) acc = gf_mul(acc, gf_load(sq, k));
        }
        gf_store(out, q, acc);
    }
}

// out[q] = out[q & 255] * hi[q >> 8]  for 256 <= q < count
__global__ void k_pow_expand(uint64_t *out, const uint64_t *hi, size_t count)
{
    for (size_t q = 256 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (size_t)gridDim.x * blockDim.x) {
        gf_store(out, q, gf_mul(gf_load(out, q & 255), gf_load(hi, q >> 8)));
    }
}

// Twiddle table in block order: entry (2^l - 1) + b, b < 2^l, is sum_k bit_{l-1-k}(b) * B_l[k] where
// B_l = the l recursed basis vectors popped by unwind level l (fft.tcc:104-110), i.e. sums[rev_l(b)]
// without the shift term.
__global__ void k_build_ltab(uint64_t *ltab, const uint64_t *rec_betas, int d, size_t count)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (((size_t)2 << l) <= e + 1) ++l;          // l = floor(log2(e + 1))
        const size_t b = e + 1 - ((size_t)1 << l);
        const int j = d - 1 - l;                        // recursion level that produced B_l
        // rec_betas[j] has d-1-j entries, levels stored back to back
        const size_t off = (size_t)j * (d - 1) - (size_t)j * (j - 1) / 2;
        gf192 acc = gf_zero();
        for (int k = 0; k < l; ++k) {
            if ((b >> (l - 1 - k)) & 1) gf_add_to(acc, gf_load(rec_betas, off + k));
        }
        gf_store(ltab, e, acc);
    }
}

// One-word numerators of the last level's twiddles, block order like ltab: entry b < 2^l is sum_k bit_{l-1-k}(b) * basis[k]
struct SmallBasis { uint32_t v[32]; };
__global__ void k_build_ltab_small(uint32_t *out, SmallBasis basis, int l, size_t count)
{
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < count; b += (size_t)gridDim.x * blockDim.x) {
        uint32_t acc = 0;
        for (int k = 0; k < l; ++k) {
            if ((b >> (l - 1 - k)) & 1) acc ^= basis.v[k];
        }
        out[b] = acc;
    }
}

struct SmallBasis64 { uint64_t v[32]; };
__global__ void k_build_ltab_small1(uint64_t *out, SmallBasis64 basis, int l, size_t count)
{
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < count; b += (size_t)gridDim.x * blockDim.x) {
        uint64_t acc = 0;
        for (int k = 0; k < l; ++k) {
            if ((b >> (l - 1 - k)) & 1) acc ^= basis.v[k];
        }
        out[b] = acc;
    }
}

__global__ void k_pad_copy(uint64_t *dst, const uint64_t *src, size_t n_src, size_t n_dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n_dst; i += (size_t)gridDim.x * blockDim.x) {
        dst[i] = i < 3 * n_src ? src[i] : 0;
    }
}

// out[i] = c for every i (degree-0 polynomial: n_coeffs <= 1)
__global__ void k_fill(uint64_t *dst, const uint64_t *src, int have_src, size_t n)
{
    const uint64_t a = have_src ? src[0] : 0, b = have_src ? src[1] : 0, c = have_src ? src[2] : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        dst[3 * i] = a; dst[3 * i + 1] = b; dst[3 * i + 2] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// phase 1: twist + Taylor-expansion network on an LDS tile
// ---------------------------------------------------------------------------------------------
struct P1Params {
    uint64_t *S;            // 2^d elements, updated in place
    const uint64_t *pow;    // power tables, level j at offset 2^(d+1) - 2^(d+1-j), 2^(d-j) entries
    int d;
    int c, h, A;            // tile = columns on bits [0,c) x rows on bits [h, h+A)
    int j0, j1;             // levels handled by this pass
    int k_start, k_end;     // first op of level j0, last op of level j1 (ops run k = d-2 .. j)
    int xcd_remap;
    int comb;               // 1: the twists whose multiplier is the same in all 64 lanes of a wavefront take the comb product
    int extra;              // >= 0: this pass also applies the twist of that level (the one after its own), see phase1_schedule
    int skip_j0;            // 1: level j0's twist was applied by the pass before
};

// the twist of one level: element gi times pow_j[gi >> j]; `uniform`: the 64 lanes of a wavefront share the multiplier — its index goes
// through the scalar unit and the product is the comb
__device__ __forceinline__ void p1_twist(uint64_t *s, int E, int tid, int nt, const P1Params &p, size_t base, int cmask, int j, bool uniform)
{
    const uint64_t *powj = p.pow + 3 * ((((size_t)2) << p.d) - (((size_t)2) << (p.d - j)));
    if (uniform) {
#pragma unroll 1
        for (int li = tid; li < E; li += nt) {
            const int li_u = (int)__builtin_amdgcn_readfirstlane((uint32_t)(li & ~63));
            const size_t gi_u = base | ((size_t)(li_u >> p.c) << p.h) | (size_t)(li_u & cmask);
            const uint64_t *t = powj + 3 * (gi_u >> j);
            const uint64_t w0 = uniform_load64(t), w1 = uniform_load64(t + 1), w2 = uniform_load64(t + 2);
            gf192 tw;
            tw.w[0] = (uint32_t)w0; tw.w[1] = (uint32_t)(w0 >> 32); tw.w[2] = (uint32_t)w1; tw.w[3] = (uint32_t)(w1 >> 32); tw.w[4] = (uint32_t)w2; tw.w[5] = (uint32_t)(w2 >> 32);
            lds_put(s, E, li, gf_mul_uniform(lds_get(s, E, li), tw));
        }
    } else {
        for (int li = tid; li < E; li += nt) {
            const size_t gi = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
            lds_put(s, E, li, gf_mul(lds_get(s, E, li), gf_load(powj, gi >> j)));
        }
    }
}

// A wavefront holds 64 consecutive tile slots: 2^c columns (bits 0..c-1 of gi) and 2^(6-c) rows (bits h..h+5-c); when all of those bits lie
// below j the 64 lanes share the multiplier of level j
__device__ __forceinline__ bool p1_twist_is_uniform(const P1Params &p, int E, int j)
{
    return p.comb && E >= 64 && p.c <= j && j >= p.h + 6 - p.c;
}

template<bool INV>
__global__ void __launch_bounds__(512) k_phase1(P1Params p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint64_t *s = iopx_smem;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int E = 1 << (p.c + p.A);
    const int midbits = p.h - p.c;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Neighbouring tiles share cache lines when a tile
    // row is narrower than a line (2^c * 24 bytes), so consecutive tiles are kept on one XCD: tile = (id % 8) * (grid / 8) + id / 8.
    const size_t o = (p.xcd_remap && (gridDim.x & 7) == 0) ? (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (size_t)blockIdx.x;
    const size_t mid = o & (((size_t)1 << midbits) - 1), hi = o >> midbits;
    const size_t base = (hi << (p.h + p.A)) | (mid << p.c);
    const int cmask = (1 << p.c) - 1;
    uint64_t *S = p.S + 3 * ((size_t)blockIdx.y << p.d);      // blockIdx.y: which vector of a batch (same domain, same tables)

    for (int li = tid; li < E; li += nt) {
        const size_t gi = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
        lds_put(s, E, li, gf_load(S, gi));
    }
    __syncthreads();

    const int jb = INV ? p.j1 : p.j0, je = INV ? p.j0 - 1 : p.j1 + 1, js = INV ? -1 : 1;
    for (int j = jb; j != je; j += js) {
        const int ks = (j == p.j0) ? p.k_start : p.d - 2;
        const int ke = (j == p.j1) ? p.k_end : j;
        const bool twist = (ks == p.d - 2) && p.pow && !(p.skip_j0 && j == p.j0);
        if (INV && p.extra >= 0 && j == p.j1 && p.pow) {           // the level undone before this one: its twist comes off here (see phase1_schedule)
            p1_twist(s, E, tid, nt, p, base, cmask, p.extra, p1_twist_is_uniform(p, E, p.extra));
            __syncthreads();
        }
        if (!INV && twist) {
            p1_twist(s, E, tid, nt, p, base, cmask, j, p1_twist_is_uniform(p, E, j));
            __syncthreads();
        }
        // forward: k = ks down to ke; inverse: k = ke up to ks
        const int nops = ks - ke + 1;
        for (int t = 0; t < nops; ++t) {
            const int k = INV ? ke + t : ks - t;
            const int kl = k - p.h + p.c;               // tile-local bit of global bit k
            for (int qd = tid; qd < (E >> 2); qd += nt) {
                const int low = qd & ((1 << kl) - 1), high = qd >> kl;
                const int b0 = (high << (kl + 2)) | low;
                const int e1 = b0 | (1 << kl), e2 = b0 | (2 << kl), e3 = b0 | (3 << kl);
                if (!INV) {
                    // S[2s+i] += S[3s+i]; S[s+i] += S[2s+i]            (fft.tcc:79-80)
                    const gf192 v2 = gf_add(lds_get(s, E, e2), lds_get(s, E, e3));
                    lds_put(s, E, e2, v2);
                    lds_put(s, E, e1, gf_add(lds_get(s, E, e1), v2));
                } else {
                    // S[q+i] += S[2q+i]; S[2q+i] += S[3q+i]            (fft.tcc:182-183)
                    const gf192 v2 = lds_get(s, E, e2);
                    lds_put(s, E, e1, gf_add(lds_get(s, E, e1), v2));
                    lds_put(s, E, e2, gf_add(v2, lds_get(s, E, e3)));
                }
            }
            __syncthreads();
        }
        if (INV && twist) {
            p1_twist(s, E, tid, nt, p, base, cmask, j, p1_twist_is_uniform(p, E, j));
            __syncthreads();
        }
        if (!INV && p.extra >= 0 && j == p.j1 && p.pow) {          // the next level's twist, while 64 consecutive elements are in one wavefront
            p1_twist(s, E, tid, nt, p, base, cmask, p.extra, p1_twist_is_uniform(p, E, p.extra));
            __syncthreads();
        }
    }

    for (int li = tid; li < E; li += nt) {
        const size_t gi = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
        gf_store(S, gi, lds_get(s, E, li));
    }
}

// ---------------------------------------------------------------------------------------------
// phase 2: butterflies in block order
// ---------------------------------------------------------------------------------------------
struct BfParams {
    const uint64_t *src;    // forward: W (2^d, shared by all cosets) when src_shared, else = dst layout
    uint64_t *dst;          // cosets * 2^d elements
    const uint64_t *ltab;   // 2^d - 1 twiddles (no shift term)
    const uint64_t *rs;     // (1 + nhi) * d shift terms: rs[v * d + l]
    const uint64_t *rs_comb;    // optional: per (local coset, level) combined shift term, [coset * d + l]
    int src_shared;
    int d, nhi;
    int c, h, A;            // upper pass tile geometry
    int p_hi, p_lo;         // pair bits handled (forward: p_hi down to p_lo)
    int a_low, c_top;       // last/first pass tile: low a_low bits x top c_top bits
    int g_bits;             // last/first pass: 2^g_bits tiles per workgroup
    size_t total_units;     // cosets (of this launch) * tiles per coset
    size_t coset_base;      // global index of the first coset of this launch (src/dst are pre-offset)
    // last level (pair bit 0) over a basis of one-word vectors ending in x^small_k: twiddle = numerator / x^small_k, the numerator the
    // XOR of ltab_small[block] and the shift numerators (rs_small[0] the shift, rs_small[1 + v] coset basis vector v); null = not used
    const uint32_t *ltab_small;
    uint32_t rs_small[25];
    int small_k;
    // the level above it (pair bit 1) over the standard basis: twiddle = two-word numerator / (x^(small1_k1 + small1_k2) (1 + x))
    const uint64_t *ltab_small1;
    uint64_t rs_small1[25];
    int small1_k1, small1_k2;
    // shift terms by byte of the (global) coset index, for transforms with too many cosets for rs_comb: entry [(g * 256 + byte) * d + l] is the
    // sum of the terms of the set bits of byte g (plus the shift's own term in group 0); null = not used
    const uint64_t *rs_tab;
    int rs_tab_groups;
    // k_bfly_edge_multi: cosets of the launch, cosets per workgroup
    size_t ncos;
    int cpw;
};

// shift term of level l's twiddles in coset `coset` of the launch
__device__ __forceinline__ gf192 bf_shift_term(const BfParams &p, size_t coset, int l)
{
    if (p.rs_comb) return gf_load(p.rs_comb, coset * p.d + l);
    const size_t gc = p.coset_base + coset;
    gf192 tw = gf_zero();
    if (p.rs_tab) {
        for (int g = 0; g < p.rs_tab_groups; ++g) gf_add_to(tw, gf_load(p.rs_tab, ((size_t)g * 256 + ((gc >> (8 * g)) & 255)) * p.d + l));
        return tw;
    }
    tw = gf_load(p.rs, (size_t)l);
    for (int v = 0; v < p.nhi; ++v) {
        if ((gc >> v) & 1) gf_add_to(tw, gf_load(p.rs, (size_t)(1 + v) * p.d + l));
    }
    return tw;
}

// twiddle of the block that contains in-coset index u at the level with pair bit pbit
__device__ __forceinline__ gf192 bf_twiddle(const BfParams &p, size_t coset, size_t u, int pbit)
{
    const int l = p.d - 1 - pbit;
    gf192 tw = gf_load(p.ltab, (((size_t)1) << l) - 1 + (u >> (pbit + 1)));
    gf_add_to(tw, bf_shift_term(p, coset, l));
    return tw;
}

// numerator of the last level's twiddle (see BfParams::ltab_small)
__device__ __forceinline__ uint32_t bf_twiddle_small(const BfParams &p, size_t coset, size_t u)
{
    uint32_t y = p.ltab_small[u >> 1] ^ p.rs_small[0];
    const size_t gc = p.coset_base + coset;
    for (int v = 0; v < p.nhi; ++v) {
        if ((gc >> v) & 1) y ^= p.rs_small[1 + v];
    }
    return y;
}

__device__ __forceinline__ uint64_t bf_twiddle_small1(const BfParams &p, size_t coset, size_t u)
{
    uint64_t y = p.ltab_small1[u >> 2] ^ p.rs_small1[0];
    const size_t gc = p.coset_base + coset;
    for (int v = 0; v < p.nhi; ++v) {
        if ((gc >> v) & 1) y ^= p.rs_small1[1 + v];
    }
    return y;
}

// the same twiddle fetched through the scalar unit: `u_uniform` is wave-uniform (64 consecutive butterflies of one block)
__device__ __forceinline__ gf192 bf_twiddle_uniform(const BfParams &p, size_t coset, size_t u_uniform, int pbit)
{
    const int l = p.d - 1 - pbit;
    const uint64_t *t = p.ltab + 3 * ((((size_t)1) << l) - 1 + (u_uniform >> (pbit + 1)));
    uint64_t w0 = uniform_load64(t), w1 = uniform_load64(t + 1), w2 = uniform_load64(t + 2);
    if (p.rs_comb) {
        const uint64_t *r = p.rs_comb + 3 * (coset * p.d + l);
        w0 ^= uniform_load64(r); w1 ^= uniform_load64(r + 1); w2 ^= uniform_load64(r + 2);
    } else {
        const uint64_t *r = p.rs + 3 * (size_t)l;
        w0 ^= uniform_load64(r); w1 ^= uniform_load64(r + 1); w2 ^= uniform_load64(r + 2);
        const size_t gc = p.coset_base + coset;
        for (int v = 0; v < p.nhi; ++v) {
            if ((gc >> v) & 1) {
                const uint64_t *q = p.rs + 3 * ((size_t)(1 + v) * p.d + l);
                w0 ^= uniform_load64(q); w1 ^= uniform_load64(q + 1); w2 ^= uniform_load64(q + 2);
            }
        }
    }
    gf192 tw;
    tw.w[0] = (uint32_t)w0; tw.w[1] = (uint32_t)(w0 >> 32); tw.w[2] = (uint32_t)w1; tw.w[3] = (uint32_t)(w1 >> 32); tw.w[4] = (uint32_t)w2; tw.w[5] = (uint32_t)(w2 >> 32);
    return tw;
}

// `uniform`: every lane of the wavefront has the same twiddle (64 consecutive butterflies of one block)
// rs_comb[c * d + l] = rs[l] + sum_{bit v of (coset_base + c)} rs[(1 + v) * d + l]
__global__ void k_rs_combine(uint64_t *out, const uint64_t *rs, int d, int nhi, size_t coset_base, size_t count)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        const size_t c = e / d, l = e % d, gc = coset_base + c;
        gf192 acc = gf_load(rs, l);
        for (int v = 0; v < nhi; ++v) {
            if ((gc >> v) & 1) gf_add_to(acc, gf_load(rs, (size_t)(1 + v) * d + l));
        }
        gf_store(out, e, acc);
    }
}

// rs_tab[(g * 256 + b) * d + l] = (g == 0 ? rs[l] : 0) + sum_{bit k of b, 8 g + k < nhi} rs[(1 + 8 g + k) * d + l]
__global__ void k_rs_tables(uint64_t *out, const uint64_t *rs, int d, int nhi, size_t count)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        const size_t l = e % d, gb = e / d, b = gb & 255, g = gb >> 8;
        gf192 acc = gf_zero();
        if (g == 0) acc = gf_load(rs, l);
        for (int k = 0; k < 8; ++k) {
            const int v = (int)(8 * g) + k;
            if (v < nhi && ((b >> k) & 1)) gf_add_to(acc, gf_load(rs, (size_t)(1 + v) * d + l));
        }
        gf_store(out, e, acc);
    }
}

// LEAN: the general product in its small-register form (gf_mul_lean), for the kernels built to run six wavefronts per SIMD
template<bool INV, bool COMB, bool LEAN = false>
__device__ __forceinline__ void bf_apply(uint64_t *s, int E, int ia, int ib, const gf192 &tw, bool uniform)
{
    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ib);
    if (!INV) {
        gf_add_to(a, (COMB && uniform) ? gf_mul_uniform(b, tw) : (LEAN ? gf_mul_lean(b, tw) : gf_mul(b, tw)));  // S[a] += S[b] * t ; S[b] += S[a]  (fft.tcc:116-117)
        gf_add_to(b, a);
    } else {
        gf_add_to(b, a);                    // S[b] += S[a] ; S[a] += S[b] * t     (fft.tcc:164-165)
        gf_add_to(a, (COMB && uniform) ? gf_mul_uniform(b, tw) : (LEAN ? gf_mul_lean(b, tw) : gf_mul(b, tw)));
    }
    lds_put(s, E, ia, a);
    lds_put(s, E, ib, b);
}

// the butterfly of the last level with the twiddle y / x^k
template<bool INV>
__device__ __forceinline__ void bf_apply_small(uint64_t *s, int E, int ia, int ib, uint32_t y, int k)
{
    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ib);
    if (!INV) {
        gf_add_to(a, gf_mul_small_over_xk(b, y, k));
        gf_add_to(b, a);
    } else {
        gf_add_to(b, a);
        gf_add_to(a, gf_mul_small_over_xk(b, y, k));
    }
    lds_put(s, E, ia, a);
    lds_put(s, E, ib, b);
}

template<bool INV>
__device__ __forceinline__ void bf_apply_small1(uint64_t *s, int E, int ia, int ib, uint64_t y, int k1, int k2)
{
    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ib);
    if (!INV) {
        gf_add_to(a, gf_mul_small2_over(b, (uint32_t)y, (uint32_t)(y >> 32), k1, k2));
        gf_add_to(b, a);
    } else {
        gf_add_to(b, a);
        gf_add_to(a, gf_mul_small2_over(b, (uint32_t)y, (uint32_t)(y >> 32), k1, k2));
    }
    lds_put(s, E, ia, a);
    lds_put(s, E, ib, b);
}

template<bool INV, bool COMB>
__global__ void __launch_bounds__(COMB ? 512 : 1024, COMB ? 6 : 1) k_bfly_upper(BfParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint64_t *s = iopx_smem;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int E = 1 << (p.c + p.A);
    const int tpc_bits = p.d - p.c - p.A;
    const size_t unit = blockIdx.x;
    const size_t coset = unit >> tpc_bits, o = unit & (((size_t)1 << tpc_bits) - 1);
    const int midbits = p.h - p.c;
    const size_t mid = o & (((size_t)1 << midbits) - 1), hi = o >> midbits;
    const size_t base = (hi << (p.h + p.A)) | (mid << p.c);
    const int cmask = (1 << p.c) - 1;
    const uint64_t *src = p.src_shared ? p.src : p.src + 3 * (coset << p.d);
    uint64_t *dst = p.dst + 3 * (coset << p.d);

    for (int li = tid; li < E; li += nt) {
        const size_t u = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
        lds_put(s, E, li, gf_load(src, u));
    }
    __syncthreads();

    const int nlev = p.p_hi - p.p_lo + 1;
    for (int t = 0; t < nlev; ++t) {
        const int pbit = INV ? p.p_lo + t : p.p_hi - t;
        const int pl = pbit - p.h + p.c;
        if (COMB) {
            // launched only for tiles with >= 64 columns (p.c >= 6), so pl >= 6 at every level: 64 consecutive butterflies share a block and the
            // twiddle is wave-uniform.  One butterfly per trip, nothing hoisted: the kernel must fit 80 VGPRs (6 waves per SIMD hide the comb
            // product's branch latency); the twiddle is fetched through a uniform index
#pragma unroll 1
            for (int bf = tid; bf < (E >> 1); bf += nt) {          // (requesting the next trip's twiddle ahead of the product changed nothing: 22.0 vs 22.1 ms)
                const int ia = ((bf >> pl) << (pl + 1)) | (bf & ((1 << pl) - 1));
                const uint32_t ia_u = __builtin_amdgcn_readfirstlane((uint32_t)(ia & ~63));
                const gf192 tw = bf_twiddle_uniform(p, coset, base | ((size_t)(ia_u >> p.c) << p.h) | (size_t)(ia_u & cmask), pbit);
                bf_apply<INV, true>(s, E, ia, ia | (1 << pl), tw, true);
            }
        } else {
            for (int bf = tid; bf < (E >> 1); bf += nt) {
                const int ia = ((bf >> pl) << (pl + 1)) | (bf & ((1 << pl) - 1));
                const gf192 tw = bf_twiddle(p, coset, base | ((size_t)(ia >> p.c) << p.h) | (size_t)(ia & cmask), pbit);
                bf_apply<INV, false>(s, E, ia, ia | (1 << pl), tw, false);
            }
        }
        __syncthreads();
    }

    for (int li = tid; li < E; li += nt) {
        const size_t u = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
        gf_store(dst, u, lds_get(s, E, li));
    }
}

// Forward: last pass — pair bits a_low-1 .. 0, then natural-order (bit-reversed) store.
// Inverse: first pass — natural-order load, pair bits 0 .. a_low-1, block-order store.
template<bool INV, bool COMB>
__global__ void __launch_bounds__(COMB ? 512 : 1024, 1) k_bfly_edge(BfParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint64_t *s = iopx_smem;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tb = p.a_low + p.c_top;                   // bits of one tile
    const int E = 1 << (tb + p.g_bits);                 // elements in LDS (2^g_bits tiles)
    const int midbits = p.d - tb;                       // tile index bits inside a coset
    const size_t unit0 = (size_t)blockIdx.x << p.g_bits;
    const int lomask = (1 << p.a_low) - 1, tmask = (1 << p.c_top) - 1;

    // natural-order side: slot sidx -> (tile g, lo, t' = rev(top)); consecutive sidx = consecutive addresses
    // block-order side  : slot e    -> (tile g, top, lo)
    if (!INV) {
        for (int e = tid; e < E; e += nt) {
            const size_t unit = unit0 + (size_t)(e >> tb);
            if (unit >= p.total_units) continue;
            const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
            const int li = e & ((1 << tb) - 1), top = li >> p.a_low, lo = li & lomask;
            const size_t u = ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo;
            const uint64_t *src = p.src_shared ? p.src : p.src + 3 * (coset << p.d);
            lds_put(s, E, e, gf_load(src, u));
        }
    } else {
        for (int sidx = tid; sidx < E; sidx += nt) {
            const size_t unit = unit0 + (size_t)(sidx >> tb);
            if (unit >= p.total_units) continue;
            const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
            const int tp = sidx & tmask, lo = (sidx >> p.c_top) & lomask;
            const int top = (int)bitrev((uint32_t)tp, p.c_top);
            const size_t v = ((size_t)bitrev((uint32_t)lo, p.a_low) << (p.d - p.a_low)) |
                             ((size_t)bitrev((uint32_t)mid, midbits) << p.c_top) | (size_t)tp;
            const int e = ((sidx >> tb) << tb) | (top << p.a_low) | lo;
            lds_put(s, E, e, gf_load(p.src + 3 * (coset << p.d), v));
        }
    }
    __syncthreads();

    for (int t = 0; t < p.a_low; ++t) {
        const int pbit = INV ? t : p.a_low - 1 - t;
        if (pbit == 0 && p.ltab_small) {                // adjacent pairs, one-word twiddle numerators
            for (int bf = tid; bf < (E >> 1); bf += nt) {
                const int ia = bf << 1;
                const size_t unit = unit0 + (size_t)(ia >> tb);
                if (unit >= p.total_units) continue;
                const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
                const int li = ia & ((1 << tb) - 1), top = li >> p.a_low, lo = li & lomask;
                const size_t u = ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo;
                bf_apply_small<INV>(s, E, ia, ia | 1, bf_twiddle_small(p, coset, u), p.small_k);
            }
            __syncthreads();
            continue;
        }
        if (pbit == 1 && p.ltab_small1) {               // two-word twiddle numerators
            for (int bf = tid; bf < (E >> 1); bf += nt) {
                const int ia = ((bf >> 1) << 2) | (bf & 1);
                const size_t unit = unit0 + (size_t)(ia >> tb);
                if (unit >= p.total_units) continue;
                const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
                const int li = ia & ((1 << tb) - 1), top = li >> p.a_low, lo = li & lomask;
                const size_t u = ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo;
                bf_apply_small1<INV>(s, E, ia, ia | 2, bf_twiddle_small1(p, coset, u), p.small1_k1, p.small1_k2);
            }
            __syncthreads();
            continue;
        }
        for (int bf = tid; bf < (E >> 1); bf += nt) {
            const int low = bf & ((1 << pbit) - 1), high = bf >> pbit;
            const int ia = (high << (pbit + 1)) | low, ib = ia | (1 << pbit);
            const size_t unit = unit0 + (size_t)(ia >> tb);
            if (unit >= p.total_units) continue;
            const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
            const int li = ia & ((1 << tb) - 1), top = li >> p.a_low, lo = li & lomask;
            const size_t u = ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo;
            bf_apply<INV, COMB>(s, E, ia, ib, bf_twiddle(p, coset, u, pbit), pbit >= 6);
        }
        __syncthreads();
    }

    if (!INV) {
        for (int sidx = tid; sidx < E; sidx += nt) {
            const size_t unit = unit0 + (size_t)(sidx >> tb);
            if (unit >= p.total_units) continue;
            const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
            const int tp = sidx & tmask, lo = (sidx >> p.c_top) & lomask;
            const int top = (int)bitrev((uint32_t)tp, p.c_top);
            const size_t v = ((size_t)bitrev((uint32_t)lo, p.a_low) << (p.d - p.a_low)) |
                             ((size_t)bitrev((uint32_t)mid, midbits) << p.c_top) | (size_t)tp;
            const int e = ((sidx >> tb) << tb) | (top << p.a_low) | lo;
            gf_store(p.dst + 3 * (coset << p.d), v, lds_get(s, E, e));
        }
    } else {
        for (int e = tid; e < E; e += nt) {
            const size_t unit = unit0 + (size_t)(e >> tb);
            if (unit >= p.total_units) continue;
            const size_t coset = unit >> midbits, mid = unit & (((size_t)1 << midbits) - 1);
            const int li = e & ((1 << tb) - 1), top = li >> p.a_low, lo = li & lomask;
            const size_t u = ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo;
            gf_store(p.dst + 3 * (coset << p.d), u, lds_get(s, E, e));
        }
    }
}


// The same pass with `cpw` cosets of one tile position per workgroup (round 5).  A twiddle is T_block + S_(coset, level): the block terms of a
// tile position are the same in every coset, so the workgroup reads them once into LDS (2^c_top (2^a_low - 1) entries, or the numerators of
// the small-numerator levels) and each coset adds its a_low shift terms — no per-butterfly twiddle load from memory (k_bfly_edge's were 41 %
// of its traffic: the table of every level re-read for each coset), and tile-local index arithmetic only.  One tile per workgroup (d >= the
// tile bits).  No register bound: the compiler takes 141 - 149 VGPRs (three wavefronts per SIMD); bounded to 128 for four, with S[a] read after
// the product so that nothing spills in the hot loop, the pass was 2.5 % slower (profiles/r05_ab_lowlive.txt).
template<bool INV>
__global__ void __launch_bounds__(256) k_bfly_edge_multi(BfParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tb = p.a_low + p.c_top;
    const int E = 1 << tb, T = 1 << p.c_top;
    const int midbits = p.d - tb;
    const size_t mid = blockIdx.x & (((size_t)1 << midbits) - 1);
    const size_t c0 = (size_t)(blockIdx.x >> midbits) * p.cpw;
    const int lomask = (1 << p.a_low) - 1, tmask = T - 1;
    const int pmin = p.ltab_small ? (p.ltab_small1 ? 2 : 1) : 0;      // general levels: pair bits pmin .. a_low - 1
    const int NTW = pmin < p.a_low ? T * ((1 << (p.a_low - pmin)) - 1) : 0;
    uint64_t *s = iopx_smem;                            // the tile: 3 E words
    uint64_t *tw = s + 3 * E;                           // block terms: 3 NTW words, level pbit at T (nblk - 1), nblk = 2^(a_low - 1 - pbit)
    uint64_t *sh = tw + 3 * NTW;                        // shift terms of the current coset: 3 a_low words, then the two small-numerator terms
    uint64_t *sm1 = sh + 3 * p.a_low + 2;               // numerators of pair bit 1: T 2^(a_low - 2) words
    uint32_t *sm0 = (uint32_t *)(sm1 + (p.ltab_small1 ? (E >> 2) : 0));    // numerators of pair bit 0: T 2^(a_low - 1) half words

    for (int pbit = pmin; pbit < p.a_low; ++pbit) {
        const int nb = p.a_low - 1 - pbit, l = p.d - 1 - pbit;
        for (int e = tid; e < (T << nb); e += nt) {
            const size_t blk = ((size_t)(e >> nb) << (p.d - p.c_top - pbit - 1)) | (mid << nb) | (size_t)(e & ((1 << nb) - 1));
            lds_put(tw, NTW, T * ((1 << nb) - 1) + e, gf_load(p.ltab, (((size_t)1) << l) - 1 + blk));
        }
    }
    if (p.ltab_small) {
        for (int e = tid; e < (E >> 1); e += nt) {
            const int top = e >> (p.a_low - 1);
            sm0[e] = p.ltab_small[((size_t)top << (p.d - p.c_top - 1)) | (mid << (p.a_low - 1)) | (size_t)(e & ((1 << (p.a_low - 1)) - 1))];
        }
    }
    if (p.ltab_small1) {
        for (int e = tid; e < (E >> 2); e += nt) {
            const int top = e >> (p.a_low - 2);
            sm1[e] = p.ltab_small1[((size_t)top << (p.d - p.c_top - 2)) | (mid << (p.a_low - 2)) | (size_t)(e & ((1 << (p.a_low - 2)) - 1))];
        }
    }

    for (int j = 0; j < p.cpw; ++j) {
        const size_t coset = c0 + j;
        if (coset >= p.ncos) break;                     // uniform over the workgroup
        const uint64_t *src = p.src_shared ? p.src : p.src + 3 * (coset << p.d);
        uint64_t *dst = p.dst + 3 * (coset << p.d);
        for (int k = pmin + tid; k <= p.a_low; k += nt) {
            if (k < p.a_low) {                          // level with pair bit k
                const gf192 t = bf_shift_term(p, coset, p.d - 1 - k);
                for (int w = 0; w < 3; ++w) sh[3 * k + w] = (uint64_t)t.w[2 * w] | ((uint64_t)t.w[2 * w + 1] << 32);
            } else if (p.ltab_small) {                  // the small-numerator levels' terms (the host fills rs_small only with the tables: nhi <= 24 then)
                const size_t gc = p.coset_base + coset;
                uint32_t y = p.rs_small[0];
                uint64_t y1 = p.ltab_small1 ? p.rs_small1[0] : 0;
                for (int v = 0; v < p.nhi && v < 24; ++v) {
                    if ((gc >> v) & 1) { y ^= p.rs_small[1 + v]; if (p.ltab_small1) y1 ^= p.rs_small1[1 + v]; }
                }
                sh[3 * p.a_low] = y;
                sh[3 * p.a_low + 1] = y1;
            }
        }
        if (!INV) {
            for (int e = tid; e < E; e += nt) {
                const int top = e >> p.a_low, lo = e & lomask;
                lds_put(s, E, e, gf_load(src, ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo));
            }
        } else {
            for (int sidx = tid; sidx < E; sidx += nt) {
                const int tp = sidx & tmask, lo = (sidx >> p.c_top) & lomask;
                const int top = (int)bitrev((uint32_t)tp, p.c_top);
                const size_t v = ((size_t)bitrev((uint32_t)lo, p.a_low) << (p.d - p.a_low)) |
                                 ((size_t)bitrev((uint32_t)mid, midbits) << p.c_top) | (size_t)tp;
                lds_put(s, E, (top << p.a_low) | lo, gf_load(src, v));
            }
        }
        __syncthreads();

        for (int t = 0; t < p.a_low; ++t) {
            const int pbit = INV ? t : p.a_low - 1 - t;
            if (pbit == 0 && p.ltab_small) {
                const uint32_t yc = (uint32_t)sh[3 * p.a_low];
                for (int bf = tid; bf < (E >> 1); bf += nt) bf_apply_small<INV>(s, E, bf << 1, (bf << 1) | 1, sm0[bf] ^ yc, p.small_k);
            } else if (pbit == 1 && p.ltab_small1) {
                const uint64_t yc = sh[3 * p.a_low + 1];
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    const int ia = ((bf >> 1) << 2) | (bf & 1);
                    bf_apply_small1<INV>(s, E, ia, ia | 2, sm1[bf >> 1] ^ yc, p.small1_k1, p.small1_k2);
                }
            } else {
                const uint64_t s0 = sh[3 * pbit], s1 = sh[3 * pbit + 1], s2 = sh[3 * pbit + 2];
                const int tbase = T * ((1 << (p.a_low - 1 - pbit)) - 1);
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    const int low = bf & ((1 << pbit) - 1), high = bf >> pbit;
                    const int ia = (high << (pbit + 1)) | low;
                    gf192 t = lds_get(tw, NTW, tbase + high);
                    t.w[0] ^= (uint32_t)s0; t.w[1] ^= (uint32_t)(s0 >> 32); t.w[2] ^= (uint32_t)s1; t.w[3] ^= (uint32_t)(s1 >> 32);
                    t.w[4] ^= (uint32_t)s2; t.w[5] ^= (uint32_t)(s2 >> 32);
                    bf_apply<INV, false, false>(s, E, ia, ia | (1 << pbit), t, false);
                }
            }
            __syncthreads();
        }

        if (!INV) {
            for (int sidx = tid; sidx < E; sidx += nt) {
                const int tp = sidx & tmask, lo = (sidx >> p.c_top) & lomask;
                const int top = (int)bitrev((uint32_t)tp, p.c_top);
                const size_t v = ((size_t)bitrev((uint32_t)lo, p.a_low) << (p.d - p.a_low)) |
                                 ((size_t)bitrev((uint32_t)mid, midbits) << p.c_top) | (size_t)tp;
                gf_store(dst, v, lds_get(s, E, (top << p.a_low) | lo));
            }
        } else {
            for (int e = tid; e < E; e += nt) {
                const int top = e >> p.a_low, lo = e & lomask;
                gf_store(dst, ((size_t)top << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)lo, lds_get(s, E, e));
            }
        }
        __syncthreads();
    }
}

// Forward last pass for a BATCH of polynomials over the same domain (the prover extends f_Az, f_Bz, f_Cz, then p_alpha' and p_alpha^ABC,
// together: r1cs_rs_iop.tcc:459-478, basic_lincheck_aux.tcc:94-118).  The twiddle of a butterfly depends on its block and coset, not on
// the polynomial, so the B * 2^pbit butterflies of one block at pair bit pbit share it: with B >= 2 the two highest edge levels (pbit 5
// and 4) fill a wavefront with one twiddle — lanes run over (polynomial, butterfly of the block) — and take the comb product instead of
// the general one (1630 against 3250 cycles per wave-butterfly; 3/4-full waves still win).  The lower levels are k_bfly_edge's.
struct BfBatchParams {
    BfParams p;
    const uint64_t *srcs[4];
    uint64_t *dsts[4];
    int batch;
};

__global__ void __launch_bounds__(512, 6) k_bfly_edge_fwd_batch(BfBatchParams q)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    const BfParams &p = q.p;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tb = p.a_low + p.c_top;                   // bits of one tile
    const int E = 1 << tb;                              // elements of ONE polynomial in LDS: one tile per workgroup (the host launches g_bits = 0)
    const int midbits = p.d - tb;
    const size_t coset = (size_t)blockIdx.x >> midbits, mid = (size_t)blockIdx.x & (((size_t)1 << midbits) - 1);
    const int lomask = (1 << p.a_low) - 1, tmask = (1 << p.c_top) - 1;
    // in-coset index of tile slot li (top, lo)
    auto index_of = [&](int li) -> size_t { return ((size_t)(li >> p.a_low) << (p.d - p.c_top)) | (mid << p.a_low) | (size_t)(li & lomask); };

    for (int b = 0; b < q.batch; ++b) {
        uint64_t *s = iopx_smem + 3 * (size_t)E * b;
        const uint64_t *src = q.srcs[b] + 3 * (coset << p.d);
        for (int e = tid; e < E; e += nt) lds_put(s, E, e, gf_load(src, index_of(e)));
    }
    __syncthreads();

    const int half = E >> 1;
    for (int t = 0; t < p.a_low; ++t) {
        const int pbit = p.a_low - 1 - t;
        const int G = 1 << pbit;                        // butterflies that share a twiddle, per polynomial
        if (pbit >= 4 && q.batch * G >= 48) {
            const int groups = half >> pbit, chunks = (q.batch * G + 63) >> 6;
            const int cshift = (chunks & (chunks - 1)) ? -1 : __builtin_ctz(chunks);       // one or two chunks with the shipped tile: no division
            // a strided loop over (work item, lane) like every other loop here; the block size is a multiple of 64, so x >> 6 is the
            // same in all lanes of a wavefront
            for (int x = tid; x < groups * chunks * 64; x += nt) {
                const int w = __builtin_amdgcn_readfirstlane(x >> 6), lane = x & 63;
                const int g = cshift >= 0 ? w >> cshift : w / chunks, ch = w - g * chunks;
                const int ia0 = g << (pbit + 1);
                const gf192 tw = bf_twiddle_uniform(p, coset, index_of(ia0), pbit);
                const int idx = ch * 64 + lane;
                if (idx < q.batch * G) {
                    const int b = idx >> pbit, ia = ia0 | (idx & (G - 1));
                    bf_apply<false, true>(iopx_smem + 3 * (size_t)E * b, E, ia, ia | G, tw, true);
                }
            }
        } else if (q.batch * G == 32 && ((half >> pbit) & 1) == 0) {
            // two blocks per wavefront, 32 lanes each (polynomial x butterfly of the block): the comb product with one multiplier per HALF of the
            // wavefront — the table of multiples built once, the window loop once per half (2775 modelled cycles against the general product's 3209)
            const int groups = half >> pbit;
            for (int x = tid; x < (groups >> 1) * 64; x += nt) {
                const int w = __builtin_amdgcn_readfirstlane(x >> 6), lane = x & 63;
                gf192 tw[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) tw[hf] = bf_twiddle_uniform(p, coset, index_of((2 * w + hf) << (pbit + 1)), pbit);
                const int g = 2 * w + (lane >> 5), idx = lane & 31;
                const int b = idx >> pbit, ia = (g << (pbit + 1)) | (idx & (G - 1)), ib = ia | G;
                uint64_t *sp = iopx_smem + 3 * (size_t)E * b;
                gf192 av = lds_get(sp, E, ia), bv = lds_get(sp, E, ib);
                gf_add_to(av, gf_mul_halves(bv, tw[0], tw[1], lane));
                gf_add_to(bv, av);
                lds_put(sp, E, ia, av); lds_put(sp, E, ib, bv);
            }
        } else if (pbit == 0 && p.ltab_small) {
            for (int x = tid; x < q.batch * half; x += nt) {
                const int b = x >> (tb - 1), ia = (x & (half - 1)) << 1;
                bf_apply_small<false>(iopx_smem + 3 * (size_t)E * b, E, ia, ia | 1, bf_twiddle_small(p, coset, index_of(ia)), p.small_k);
            }
        } else if (pbit == 1 && p.ltab_small1) {
            for (int x = tid; x < q.batch * half; x += nt) {
                const int b = x >> (tb - 1), bf = x & (half - 1);
                const int ia = ((bf >> 1) << 2) | (bf & 1);
                bf_apply_small1<false>(iopx_smem + 3 * (size_t)E * b, E, ia, ia | 2, bf_twiddle_small1(p, coset, index_of(ia)), p.small1_k1, p.small1_k2);
            }
        } else {
            for (int x = tid; x < q.batch * half; x += nt) {
                const int b = x >> (tb - 1), bf = x & (half - 1);
                const int low = bf & (G - 1), high = bf >> pbit;
                const int ia = (high << (pbit + 1)) | low;
                bf_apply<false, false, true>(iopx_smem + 3 * (size_t)E * b, E, ia, ia | G, bf_twiddle(p, coset, index_of(ia), pbit), false);
            }
        }
        __syncthreads();
    }

    for (int b = 0; b < q.batch; ++b) {
        const uint64_t *s = iopx_smem + 3 * (size_t)E * b;
        uint64_t *dst = q.dsts[b] + 3 * (coset << p.d);
        for (int sidx = tid; sidx < E; sidx += nt) {
            const int tp = sidx & tmask, lo = (sidx >> p.c_top) & lomask;
            const int top = (int)bitrev((uint32_t)tp, p.c_top);
            const size_t v = ((size_t)bitrev((uint32_t)lo, p.a_low) << (p.d - p.a_low)) |
                             ((size_t)bitrev((uint32_t)mid, midbits) << p.c_top) | (size_t)tp;
            gf_store(dst, v, lds_get(s, E, (top << p.a_low) | lo));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: plans
// ---------------------------------------------------------------------------------------------
struct AddPlan {
    int d = 0;
    std::vector<hgf192> basis;                  // basis[0..d)
    std::vector<hgf192> beta, betainv;          // per level j
    DevBuf ltab;                                // 2^d - 1 twiddles
    DevBuf pow_fwd, pow_inv;                    // 2^(d+1) - 2 entries each, built on first use
    bool have_fwd = false, have_inv = false;
    DevBuf rs;                                  // per-call shift terms (stream ordered) of a (shift, coset basis) the cache below does not hold
    size_t rs_cap = 0;
    // The shift terms are a function of (shift, coset basis vectors) only, and a prover asks for the same few per domain in every proof: kept on the
    // device per pair (host: (1 + nhi) d recursions; device: one constant-carrying launch — both off the path after the first proof).
    std::map<std::vector<uint64_t>, std::unique_ptr<DevBuf>> rs_cache;
    const uint64_t *rs_cur = nullptr;           // the terms the next launches read: a cache entry or `rs`
    // one-word basis ending in a power of x (the standard basis): numerators of the last level's twiddles, see BfParams::ltab_small
    DevBuf ltab_small;
    int small_k = -1;                           // basis[d-1] = x^small_k, or -1
    uint32_t rs_small[25] = { 0 };              // per call, like rs
    bool rs_small_ok = false;
    // ... and its second-to-last vector such that the next recursed basis ends in x^small1_e (1 + x) / x^(2 small_k) (the standard
    // basis): two-word numerators of the second-to-last level, see BfParams::ltab_small1
    DevBuf ltab_small1;
    int small1_e = -1;
    uint64_t rs_small1[25] = { 0 };

    // recursed shift of an arbitrary element: GF(2)-linear in s (fft.tcc:93-95 / :153-154)
    void recursed_shifts(const hgf192 &s, hgf192 *out_by_unwind_level) const
    {
        hgf192 s2 = s;
        for (int j = 0; j < d; ++j) {
            const hgf192 ns = s2 * betainv[j];
            out_by_unwind_level[d - 1 - j] = ns;        // unwind level l pops recursed_shifts[d-1-l]
            s2 = ns.squared() + ns;
        }
    }
};

static bool one_word(const hgf192 &e) { return e.w[1] == 0 && e.w[2] == 0 && (e.w[0] >> 32) == 0; }

// n^2 + n x^k as a polynomial (k < 32): the numerator over x^(2k) of (n / x^k)^2 + n / x^k
static uint64_t recursed_numerator(uint32_t n, int k)
{
    uint64_t sq = 0;
    for (int i = 0; i < 32; ++i) sq |= (uint64_t)((n >> i) & 1) << (2 * i);
    return sq ^ ((uint64_t)n << k);
}

static std::mutex g_plan_mu;
static std::map<std::vector<uint64_t>, std::unique_ptr<AddPlan>> g_plans;

static int grid_for(size_t work, int threads)
{
    size_t g = (work + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

static int build_pow_table(uint64_t *out, const uint64_t *d_sq, int nb)
{
    const size_t count = (size_t)1 << nb;
    if (nb <= 8) {
        { ProfScope ps_("k_pow_direct"); hipLaunchKernelGGL(k_pow_direct, dim3(grid_for(count, 256)), dim3(256), 0, stream(), out, d_sq, nb, count); }
        return IOPX_OK;
    }
    TmpBuf tmp;                                         // pooled: recycled in stream order, no hipFree and no drain of the stream per table
    int rc = tmp.alloc(((size_t)1 << (nb - 8)) * 24);
    if (rc != IOPX_OK) return rc;
    rc = build_pow_table(tmp.u64(), d_sq + 3 * 8, nb - 8);
    if (rc != IOPX_OK) return rc;
    { ProfScope ps_("k_pow_direct"); hipLaunchKernelGGL(k_pow_direct, dim3(1), dim3(256), 0, stream(), out, d_sq, 8, (size_t)256); }
    { ProfScope ps_("k_pow_expand"); hipLaunchKernelGGL(k_pow_expand, dim3(grid_for(count - 256, 256)), dim3(256), 0, stream(), out, (const uint64_t *)tmp.u64(), count); }
    return IOPX_OK;
}

static int build_pow_tables(AddPlan &pl, bool inverse)
{
    ColdScope cold_("additive FFT twist-power tables");
    const int d = pl.d;
    DevBuf &buf = inverse ? pl.pow_inv : pl.pow_fwd;
    int rc = buf.alloc(((((size_t)2) << d) - 2) * 24);
    if (rc != IOPX_OK) return rc;
    // squares beta_j^(2^k), k < d - j, all levels back to back
    std::vector<uint64_t> sq;
    std::vector<size_t> off(d);
    for (int j = 0; j < d; ++j) {
        off[j] = sq.size();
        hgf192 x = inverse ? pl.betainv[j] : pl.beta[j];
        for (int k = 0; k < d - j; ++k) {
            sq.insert(sq.end(), x.w, x.w + 3);
            x = x.squared();
        }
    }
    DevBuf dsq;
    rc = dsq.alloc(sq.size() * 8);
    if (rc != IOPX_OK) return rc;
    { int urc_ = upload(dsq.p, sq.data(), sq.size() * 8); if (urc_ != IOPX_OK) return urc_; }
    for (int j = 0; j < d; ++j) {
        uint64_t *out = buf.u64() + 3 * ((((size_t)2) << d) - (((size_t)2) << (d - j)));
        rc = build_pow_table(out, dsq.u64() + off[j], d - j);
        if (rc != IOPX_OK) return rc;
    }
    IOPX_HIP(hipStreamSynchronize(stream()));
    (inverse ? pl.have_inv : pl.have_fwd) = true;
    return IOPX_OK;
}

static int get_plan(const uint64_t *basis, int d, AddPlan **out)
{
    std::vector<uint64_t> key(basis, basis + 3 * (size_t)d);
    key.push_back((uint64_t)d);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) { *out = it->second.get(); return IOPX_OK; }

    ColdScope cold_("additive FFT plan");
    std::unique_ptr<AddPlan> pl(new AddPlan());
    pl->d = d;
    for (int i = 0; i < d; ++i) pl->basis.push_back(hgf192::from_words(basis + 3 * i));
    // recursed bases (fft.tcc:55-96); rec[j] = newbetas of level j (d-1-j entries)
    std::vector<hgf192> betas2(pl->basis);
    std::vector<uint64_t> rec_flat;
    for (int j = 0; j < d; ++j) {
        const hgf192 beta = betas2[d - 1 - j];
        if (beta.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "additive FFT: basis vectors are linearly dependent");
        const hgf192 binv = beta.inverse();
        pl->beta.push_back(beta);
        pl->betainv.push_back(binv);
        for (int i = 0; i < d - 1 - j; ++i) {
            const hgf192 nb = betas2[i] * binv;
            rec_flat.insert(rec_flat.end(), nb.w, nb.w + 3);
            betas2[i] = nb.squared() + nb;
        }
    }
    if (d >= 1) {
        const size_t count = ((size_t)1 << d) - 1;
        int rc = pl->ltab.alloc(count * 24);
        if (rc != IOPX_OK) return rc;
        DevBuf drec;
        rc = drec.alloc(rec_flat.size() * 8);
        if (rc != IOPX_OK) return rc;
        if (!rec_flat.empty()) { int urc_ = upload(drec.p, rec_flat.data(), rec_flat.size() * 8); if (urc_ != IOPX_OK) return urc_; }
        { ProfScope ps_("k_build_ltab"); hipLaunchKernelGGL(k_build_ltab, dim3(grid_for(count, 256)), dim3(256), 0, stream(), pl->ltab.u64(), (const uint64_t *)drec.u64(), d, count); }
        // the last level's numerators when every basis vector is one word and the last one a power of x
        bool small = d >= 2 && d <= 32 && tuning().small_last;
        for (int i = 0; small && i < d; ++i) small = one_word(pl->basis[i]);
        const uint64_t last = pl->basis[d - 1].w[0];
        if (small && (last & (last - 1)) == 0) {
            SmallBasis sb;
            memset(&sb, 0, sizeof(sb));
            for (int i = 0; i < d - 1; ++i) sb.v[i] = (uint32_t)pl->basis[i].w[0];
            const size_t n_small = (size_t)1 << (d - 1);
            rc = pl->ltab_small.alloc(n_small * 4);
            if (rc != IOPX_OK) return rc;
            hipLaunchKernelGGL(k_build_ltab_small, dim3(grid_for(n_small, 256)), dim3(256), 0, stream(), (uint32_t *)pl->ltab_small.p, sb, d - 1, n_small);
            pl->small_k = __builtin_ctzll(last);
            // next level: vector n recurses to (n^2 + n x^k) / x^(2k); the level normalises by the last of them
            if (d >= 3 && pl->small_k < 32) {
                const uint64_t N = recursed_numerator((uint32_t)pl->basis[d - 2].w[0], pl->small_k);
                if (N != 0 && (N >> __builtin_ctzll(N)) == 3) {
                    SmallBasis64 sb1;
                    memset(&sb1, 0, sizeof(sb1));
                    for (int i = 0; i < d - 2; ++i) sb1.v[i] = recursed_numerator((uint32_t)pl->basis[i].w[0], pl->small_k);
                    const size_t n1 = (size_t)1 << (d - 2);
                    rc = pl->ltab_small1.alloc(n1 * 8);
                    if (rc != IOPX_OK) return rc;
                    hipLaunchKernelGGL(k_build_ltab_small1, dim3(grid_for(n1, 256)), dim3(256), 0, stream(), pl->ltab_small1.u64(), sb1, d - 2, n1);
                    pl->small1_e = __builtin_ctzll(N);
                }
            }
        }
        IOPX_HIP(hipStreamSynchronize(stream()));
    }
    *out = pl.get();
    g_plans[key] = std::move(pl);
    return IOPX_OK;
}

static int upload_rs(AddPlan &pl, const hgf192 &shift, const uint64_t *hi_basis, int nhi)
{
    const int d = pl.d;
    pl.rs_small_ok = pl.small_k >= 0 && nhi <= 24 && one_word(shift);
    if (pl.rs_small_ok) {
        pl.rs_small[0] = (uint32_t)shift.w[0];
        for (int v = 0; v < nhi && pl.rs_small_ok; ++v) {
            const hgf192 e = hgf192::from_words(hi_basis + 3 * v);
            pl.rs_small_ok = one_word(e);
            pl.rs_small[1 + v] = (uint32_t)e.w[0];
        }
        for (int v = 0; v <= nhi; ++v) pl.rs_small1[v] = recursed_numerator(pl.rs_small[v], pl.small_k);
    }
    std::vector<uint64_t> key(shift.w, shift.w + 3);
    key.push_back((uint64_t)nhi);
    if (nhi) key.insert(key.end(), hi_basis, hi_basis + 3 * (size_t)nhi);
    auto hit = pl.rs_cache.find(key);
    if (hit != pl.rs_cache.end()) { pl.rs_cur = hit->second->u64(); return IOPX_OK; }
    std::vector<hgf192> rs((size_t)(1 + nhi) * d);
    pl.recursed_shifts(shift, rs.data());
    for (int v = 0; v < nhi; ++v) pl.recursed_shifts(hgf192::from_words(hi_basis + 3 * v), rs.data() + (size_t)(1 + v) * d);
    const size_t bytes = rs.size() * 24;
    if (bytes && pl.rs_cache.size() < 64) {      // a prover's domains are few; anything beyond that takes the shared per-call buffer
        std::unique_ptr<DevBuf> buf(new DevBuf());
        int rc = buf->alloc(bytes);
        if (rc != IOPX_OK) return rc;
        if ((rc = upload(buf->p, rs.data(), bytes)) != IOPX_OK) return rc;
        pl.rs_cur = buf->u64();
        pl.rs_cache.emplace(std::move(key), std::move(buf));
        return IOPX_OK;
    }
    if (bytes > pl.rs_cap) {
        IOPX_HIP(hipStreamSynchronize(stream()));
        int rc = pl.rs.alloc(bytes);
        if (rc != IOPX_OK) return rc;
        pl.rs_cap = bytes;
    }
    if (bytes) { int urc_ = upload(pl.rs.p, rs.data(), bytes); if (urc_ != IOPX_OK) return urc_; }
    pl.rs_cur = pl.rs.u64();
    return IOPX_OK;
}

template<typename K>
static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) IOPX_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return IOPX_OK;
}

// ---- phase-1 schedule --------------------------------------------------------------------------
struct P1Pass { int c, h, A, j0, j1, k_start, k_end, extra, skip_j0; };

static std::vector<P1Pass> phase1_schedule(int d)
{
    std::vector<P1Pass> sched;
    if (d <= P1_TILE_BITS) {
        sched.push_back({0, 0, d, 0, d - 1, d - 2, d - 1, -1, 0});
        return sched;
    }
    const int A = P1_TILE_BITS - P1_COLS;
    const int fin_cols = d <= tuning().p1_fin_tile_bits ? 0 : tuning().p1_fin_cols;
    const int Afin = (d <= tuning().p1_fin_tile_bits ? d : tuning().p1_fin_tile_bits) - fin_cols;
    const int hfin = d - Afin;              // levels >= hfin live entirely in the top Afin bits
    for (int j = 0; j < hfin; ++j) {
        int k = d - 2;
        while (k >= j) {
            if (k + 1 <= P1_TILE_BITS - 1) {   // the rest of this level fits a contiguous tile
                sched.push_back({0, 0, P1_TILE_BITS, j, j, k, j, -1, 0});
                break;
            }
            const int h = k + 2 - A;
            const int ke = h > j ? h : j;
            sched.push_back({P1_COLS, h, A, j, j, k, ke, -1, 0});
            k = ke - 1;
        }
    }
    sched.push_back({fin_cols, hfin, Afin, hfin, d - 1, d - 2, d - 1, -1, 0});
    // The twist of level j multiplies element i by a power indexed by i >> j: from level 6 on, 64 consecutive elements share it.  The pass
    // that twists level j holds 8-element runs (its tile wants rows on high bits), but the last pass of level j - 1 holds a contiguous
    // tile: when there is one, level j's twist moves there — wave-uniform multipliers, comb product — and the pass after skips it.
    // (Forward: after that pass's own steps; inverse: before them.)
    if (tuning().comb && tuning().p1_comb) {
        for (size_t i = 0; i + 1 < sched.size(); ++i) {
            P1Pass &a = sched[i], &b = sched[i + 1];
            const bool contiguous_last = a.c == 0 && a.h == 0 && a.j0 == a.j1 && b.j0 == a.j0 + 1 && b.k_start == d - 2;
            if (contiguous_last && b.j0 >= 6) { a.extra = b.j0; b.skip_j0 = 1; }
        }
    }
    return sched;
}

// One phase-1 launch.  d_eff / pow: the transform the pass belongs to — the plan's own (d, table base) or, for a residue class of the
// coefficient index (run_phase1 below), the sub-transform of dimension d - r whose tables are the plan's from level r on.
template<bool INV>
static int launch_phase1_pass(const P1Pass &ps, uint64_t *S, const uint64_t *pow, int d_eff, size_t batch)
{
    P1Params p;
    p.S = S;
    p.pow = pow;
    p.d = d_eff; p.c = ps.c; p.h = ps.h; p.A = ps.A;
    p.j0 = ps.j0; p.j1 = ps.j1; p.k_start = ps.k_start; p.k_end = ps.k_end;
    p.xcd_remap = (1);
    p.comb = tuning().comb && tuning().p1_comb;
    p.extra = ps.extra; p.skip_j0 = ps.skip_j0;
    const int tbits = ps.c + ps.A;
    const size_t lds = ((size_t)24) << tbits;
    const size_t blocks = (size_t)1 << (d_eff - tbits);
    const int threads = (1 << tbits) >= 4 * BLOCK_THREADS ? BLOCK_THREADS : ((1 << tbits) >= 256 ? 64 * ((1 << tbits) / 256) : 64);
    int rc = set_lds(k_phase1<INV>, lds);
    if (rc != IOPX_OK) return rc;
    static char names[64][2][40];
    char *nm = names[ps.j0 & 63][ps.k_start == d_eff - 2 ? 1 : 0];
    if (!nm[0]) snprintf(nm, 40, opt("IOPX_PROFILE_LEVELS", 0) ? "k_phase1_%s_L%02d_%s" : "k_phase1_%s", INV ? "inv" : "fwd", ps.j0, ps.k_start == d_eff - 2 ? "tw" : "x");
    { ProfScope ps_(nm, (batch << d_eff) * 48); hipLaunchKernelGGL(k_phase1<INV>, dim3((unsigned)blocks, (unsigned)batch), dim3(threads), lds, stream(), p); }
    return IOPX_OK;
}

// dst[b][l] = src[b][(l << r) | cls]  (the residue class cls of the index mod 2^r, as a contiguous vector); batch vectors of 2^d / 2^(d-r) elements
__global__ void k_class_extract(const uint64_t *src, uint64_t *dst, int d, int r, size_t cls, size_t batch)
{
    const size_t per = (size_t)1 << (d - r), total = batch * per * 3;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t w = t % 3, e = t / 3, l = e & (per - 1), b = e >> (d - r);
        dst[t] = src[3 * ((b << d) | (l << r) | cls) + w];
    }
}

// dst[b][(l << r) | q] = src[q][b][l]  (an all-gather's output, rank-major, back to index order)
__global__ void k_class_merge(const uint64_t *src, uint64_t *dst, int d, int r, size_t batch)
{
    const size_t per = (size_t)1 << (d - r), n = (size_t)1 << d, total = batch * n * 3;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t w = t % 3, e = t / 3, i = e & (n - 1), b = e >> d, q = i & (((size_t)1 << r) - 1), l = i >> r;
        dst[t] = src[3 * ((q * batch + b) * per + l) + w];
    }
}

extern "C" int iopx_comm_all_gather_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);

// `batch` vectors of 2^d elements stored back to back share every launch (grid.y): small transforms are latency-bound per
// pass, so a batch costs little more than one.
//
// With a communicator of N = 2^r ranks bound for transforms (iopx_comm_bind_transforms) and d large enough, the levels are split over the
// ranks: level j multiplies element i by a power indexed by i >> j and its Taylor operations pair index bits k, k + 1 for k >= j, so from
// level r on elements of different residues i mod N never meet.  Levels < r run on the whole vector on every rank; then rank q runs levels
// r .. d - 1 on the residue class q — as a contiguous vector it is phase 1 of a 2^(d-r)-point transform whose level-j' table is the plan's
// level r + j' table — and an all-gather brings the classes back (inverse: the other way round).  Every rank must hold the same input.
template<bool INV>
static int run_phase1(AddPlan &pl, uint64_t *S, size_t batch = 1)
{
    const int d = pl.d;
    if (d == 0) return IOPX_OK;
    if (!(INV ? pl.have_inv : pl.have_fwd)) {
        int rc = build_pow_tables(pl, INV);
        if (rc != IOPX_OK) return rc;
    }
    const uint64_t *pow = INV ? pl.pow_inv.u64() : pl.pow_fwd.u64();
    std::vector<P1Pass> sched = phase1_schedule(d);
    const int n = (int)sched.size();
    const CommInfo ci = transform_comm();
    int r = 0;
    while ((1 << r) < ci.world) ++r;
    // the head (levels < r) must be whole passes of the schedule: true while each of those levels has passes of its own
    bool split = ci.comm && ci.world > 1 && d >= env_int("IOPX_P1_SHARD_MIN_D", 16, 2, 40) && r < 6 && d - r >= 2;
    for (const P1Pass &ps : sched) if (ps.j0 < r && ps.j1 >= r) split = false;
    if (!split) {
        for (int i = 0; i < n; ++i) {
            int rc = launch_phase1_pass<INV>(sched[INV ? n - 1 - i : i], S, pow, d, batch);
            if (rc != IOPX_OK) return rc;
        }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    }
    const int dt = d - r;
    const size_t per = (size_t)1 << dt;
    const uint64_t *pow_tail = pow + 3 * ((((size_t)2) << d) - (((size_t)2) << dt));
    const std::vector<P1Pass> tail = phase1_schedule(dt);
    TmpBuf cls, all;
    int rc = cls.alloc(batch * per * 24);
    if (rc != IOPX_OK) return rc;
    rc = all.alloc((batch * 24) << d);
    if (rc != IOPX_OK) return rc;
    auto head = [&]() -> int {
        for (int i = 0; i < n; ++i) {
            const P1Pass &ps = sched[INV ? n - 1 - i : i];
            if (ps.j0 >= r) continue;
            int hrc = launch_phase1_pass<INV>(ps, S, pow, d, batch);
            if (hrc != IOPX_OK) return hrc;
        }
        return IOPX_OK;
    };
    if (!INV && (rc = head()) != IOPX_OK) return rc;
    { ProfScope ps_("k_class_extract", batch * per * 48); hipLaunchKernelGGL(k_class_extract, dim3(grid_for(batch * per * 3, 256)), dim3(256), 0, stream(), (const uint64_t *)S, cls.u64(), d, r, (size_t)ci.rank, batch); }
    const int nt = (int)tail.size();
    for (int i = 0; i < nt; ++i) {
        rc = launch_phase1_pass<INV>(tail[INV ? nt - 1 - i : i], cls.u64(), pow_tail, dt, batch);
        if (rc != IOPX_OK) return rc;
    }
    rc = iopx_comm_all_gather_dev(ci.comm, cls.p, all.p, batch * per * 24);
    if (rc != IOPX_OK) return rc;
    { ProfScope ps_("k_class_merge", (batch << d) * 48); hipLaunchKernelGGL(k_class_merge, dim3(grid_for((batch * 3) << d, 256)), dim3(256), 0, stream(), (const uint64_t *)all.u64(), S, d, r, batch); }
    if (INV && (rc = head()) != IOPX_OK) return rc;
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// ---- phase-2 schedule --------------------------------------------------------------------------
struct P2Geom { int a_low, c_top; };

#define EDGE_TILE_BITS (tuning().edge_tile_bits)
static P2Geom phase2_geom(int d)
{
    if (d < EDGE_TILE_BITS) return {d, 0};
    return {EDGE_TILE_BITS - P2_TOP, P2_TOP};
}

static void set_small_last(BfParams &p, const AddPlan &pl)
{
    if (!pl.rs_small_ok) return;            // p was zeroed: ltab_small stays null
    p.ltab_small = (const uint32_t *)pl.ltab_small.p;
    memcpy(p.rs_small, pl.rs_small, sizeof(p.rs_small));
    p.small_k = pl.small_k;
    if (pl.small1_e >= 0) {
        p.ltab_small1 = pl.ltab_small1.u64();
        memcpy(p.rs_small1, pl.rs_small1, sizeof(p.rs_small1));
        p.small1_k1 = pl.small1_e >> 1;
        p.small1_k2 = pl.small1_e - p.small1_k1;
    }
}

// forward: W (2^d, block order after phase 1) -> dst (2^nhi cosets * 2^d, natural order);
// inverse: single coset, natural-order src -> block-order dst (src != dst).
// The edge pass permutes (bit reversal), so it never runs in place across workgroups: the forward
// transform keeps the upper passes in W (one coset) or in a scratch buffer holding a group of cosets.
template<bool INV>
static int run_phase2(AddPlan &pl, const uint64_t *src, uint64_t *dst, int nhi, size_t coset_begin = 0, size_t coset_count = 0, bool upper_only = false,
                      const uint64_t *shared_rs_comb = nullptr)
{
    const int d = pl.d;
    const P2Geom g = phase2_geom(d);
    const size_t cosets = coset_count ? coset_count : ((size_t)1 << nhi);
    const size_t nd = (size_t)1 << d;
    BfParams p;
    memset(&p, 0, sizeof(p));
    p.ltab = pl.ltab.u64();
    p.rs = pl.rs_cur;
    p.d = d; p.nhi = nhi;
    p.a_low = g.a_low; p.c_top = g.c_top;
    set_small_last(p, pl);
    // combined per-coset shift terms (one load per twiddle instead of 1 + nhi); skipped for huge coset counts
    TmpBuf rs_comb;
    size_t comb_base = 0, comb_count = 0;
    const uint64_t *rs_comb_ptr = shared_rs_comb;       // the caller's table starts at coset_begin
    if (shared_rs_comb) {
        comb_base = coset_begin; comb_count = cosets;
    } else if (!INV && nhi > 0 && cosets * (size_t)d <= ((size_t)1 << tuning().rs_comb_cap_log2)) {
        int rcc = rs_comb.alloc(cosets * d * 24);
        if (rcc != IOPX_OK) return rcc;
        comb_base = coset_begin; comb_count = cosets;
        { ProfScope ps_("k_rs_combine"); hipLaunchKernelGGL(k_rs_combine, dim3(grid_for(cosets * d, 256)), dim3(256), 0, stream(), rs_comb.u64(), pl.rs_cur, d, nhi, coset_begin, cosets * (size_t)d); }
        rs_comb_ptr = rs_comb.u64();
    }
    // too many cosets for that table (a short polynomial over a long domain: the prover's f_1v, 16 coefficients over 2^25 points): shift terms
    // by byte of the coset index, 1 + nhi / 8 loads per twiddle instead of 1 + nhi conditional ones
    TmpBuf rs_tab;
    if (!INV && nhi > 0 && !comb_count) {
        const int groups = (nhi + 7) / 8;
        const size_t count = (size_t)groups * 256 * d;
        int rct = rs_tab.alloc(count * 24);
        if (rct != IOPX_OK) return rct;
        { ProfScope ps_("k_rs_combine"); hipLaunchKernelGGL(k_rs_tables, dim3(grid_for(count, 256)), dim3(256), 0, stream(), rs_tab.u64(), pl.rs_cur, d, nhi, count); }
        p.rs_tab = rs_tab.u64();
        p.rs_tab_groups = groups;
    }

    // upper passes over pair bits [a_low, d): chunks of up to A bits, from the top (forward order)
    struct Up { int h, A, c; };
    std::vector<Up> ups;
    {
        int top = d;                        // exclusive
        while (top > g.a_low) {
            int A = TILE_BITS - P2_COLS;
            if (top - A < g.a_low) A = top - g.a_low;
            const int c = TILE_BITS - A > g.a_low ? g.a_low : TILE_BITS - A;
            ups.push_back({top - A, A, c});
            top -= A;
        }
    }
    const int tb = g.a_low + g.c_top;

    auto launch_edge = [&](const uint64_t *s, uint64_t *dd, int shared, size_t ncos, size_t cbase) -> int {
        int g_bits = EDGE_TILE_BITS - tb;
        const size_t units = ncos << (d - tb);
        while (g_bits > 0 && ((size_t)1 << g_bits) > units) --g_bits;
        const size_t lds = ((size_t)24) << (tb + g_bits);
        const size_t blocks = (units + ((size_t)1 << g_bits) - 1) >> g_bits;
        const int elems = 1 << (tb + g_bits);
        const int maxt = (256);
        const int threads = elems >= 4 * maxt ? maxt : (elems >= 256 ? elems / 4 : 64);
        p.src = s; p.dst = dd; p.src_shared = shared;
        p.g_bits = g_bits; p.total_units = units; p.coset_base = cbase;
        p.rs_comb = comb_count ? rs_comb_ptr + 3 * (cbase - comb_base) * d : nullptr;
        int rc;
        const int pmin = p.ltab_small ? (p.ltab_small1 ? 2 : 1) : 0;
        const size_t ntw = pmin < p.a_low ? ((size_t)1 << p.c_top) * (((size_t)1 << (p.a_low - pmin)) - 1) : 0;
        const size_t lds_m = lds + 24 * ntw + 8 * (3 * (size_t)p.a_low + 2) + (p.ltab_small1 ? 8 * ((size_t)elems >> 2) : 0) + (p.ltab_small ? 4 * ((size_t)elems >> 1) : 0);
        if (tuning().edge_multi > 0 && g_bits == 0 && tb == EDGE_TILE_BITS && threads <= 256 && lds_m <= 80 * 1024) {      // (two workgroups per CU at least)
            size_t cpw = (size_t)tuning().edge_multi < ncos ? (size_t)tuning().edge_multi : ncos;
            cpw = (ncos + (ncos + cpw - 1) / cpw - 1) / ((ncos + cpw - 1) / cpw);           // even shares
            p.ncos = ncos; p.cpw = (int)cpw;
            const size_t blocks_m = ((ncos + cpw - 1) / cpw) << (d - tb);
            if ((rc = set_lds(k_bfly_edge_multi<INV>, lds_m)) != IOPX_OK) return rc;
            { ProfScope ps_(INV ? "k_bfly_edge_inv" : "k_bfly_edge_fwd", (ncos << d) * 48, ((ncos << d) >> 1) * (size_t)p.a_low); hipLaunchKernelGGL((k_bfly_edge_multi<INV>), dim3((unsigned)blocks_m), dim3(threads), lds_m, stream(), p); }
            return IOPX_OK;
        }
        if ((rc = set_lds(k_bfly_edge<INV, false>, lds)) != IOPX_OK) return rc;
        { ProfScope ps_(INV ? "k_bfly_edge_inv" : "k_bfly_edge_fwd", (ncos << d) * 48, ((ncos << d) >> 1) * (size_t)p.a_low); hipLaunchKernelGGL((k_bfly_edge<INV, false>), dim3((unsigned)blocks), dim3(threads), lds, stream(), p); }
        return IOPX_OK;
    };
    auto launch_upper = [&](const Up &u, const uint64_t *s, uint64_t *dd, int shared, size_t ncos, size_t cbase) -> int {
        p.src = s; p.dst = dd; p.src_shared = shared;
        p.c = u.c; p.h = u.h; p.A = u.A;
        p.p_hi = u.h + u.A - 1; p.p_lo = u.h;
        const int tbits = u.c + u.A;
        const size_t lds = ((size_t)24) << tbits;
        const size_t blocks = ncos << (d - tbits);
        p.total_units = blocks; p.coset_base = cbase;
        p.rs_comb = comb_count ? rs_comb_ptr + 3 * (cbase - comb_base) * d : nullptr;
        const int maxt = (tuning().comb && u.c < 6) ? 1024 : tuning().p2_threads;
        int threads = (1 << tbits) >= 2 * maxt ? maxt : ((1 << tbits) >= 128 ? (1 << tbits) / 2 : 64);
        int rc;
        if (tuning().comb && u.c >= 6) {
            if ((rc = set_lds(k_bfly_upper<INV, true>, lds)) != IOPX_OK) return rc;
            { ProfScope ps_(INV ? "k_bfly_upper_inv" : "k_bfly_upper_fwd", (ncos << d) * 48, ((ncos << d) >> 1) * (size_t)u.A); hipLaunchKernelGGL((k_bfly_upper<INV, true>), dim3((unsigned)blocks), dim3(threads), lds, stream(), p); }
        } else {
            if ((rc = set_lds(k_bfly_upper<INV, false>, lds)) != IOPX_OK) return rc;
            { ProfScope ps_(INV ? "k_bfly_upper_inv" : "k_bfly_upper_fwd", (ncos << d) * 48, ((ncos << d) >> 1) * (size_t)u.A); hipLaunchKernelGGL((k_bfly_upper<INV, false>), dim3((unsigned)blocks), dim3(threads), lds, stream(), p); }
        }
        return IOPX_OK;
    };

    int rc;
    if (INV) {
        // nhi = 0: the "cosets" of the unit index are the vectors of a batch (identical twiddles)
        rc = launch_edge(src, dst, 0, cosets, 0);
        if (rc != IOPX_OK) return rc;
        for (size_t i = ups.size(); i-- > 0; ) {
            rc = launch_upper(ups[i], dst, dst, 0, cosets, 0);
            if (rc != IOPX_OK) return rc;
        }
    } else if (ups.empty()) {
        rc = launch_edge(src, dst, 1, cosets, coset_begin);
        if (rc != IOPX_OK) return rc;
    } else if (nhi == 0) {
        uint64_t *W = const_cast<uint64_t *>(src);
        for (size_t i = 0; i < ups.size(); ++i) {
            rc = launch_upper(ups[i], W, W, 1, 1, 0);
            if (rc != IOPX_OK) return rc;
        }
        rc = launch_edge(W, dst, 1, 1, 0);
        if (rc != IOPX_OK) return rc;
    } else if (upper_only) {
        // the caller's dst is the scratch of a batched last pass: `cosets` cosets in block order
        for (size_t i = 0; i < ups.size(); ++i) {
            rc = launch_upper(ups[i], i == 0 ? src : dst, dst, i == 0 ? 1 : 0, cosets, coset_begin);
            if (rc != IOPX_OK) return rc;
        }
    } else {
        size_t group = SCRATCH_BYTES / (nd * 24);
        if (group < 1) group = 1;
        if (group > cosets) group = cosets;
        group = (cosets + (cosets + group - 1) / group - 1) / ((cosets + group - 1) / group);     // even groups: 31 cosets go as 8, 8, 8, 7, not 10, 10, 10, 1 (a one-coset launch leaves most CUs idle in its tail)
        TmpBuf scratch;
        rc = scratch.alloc(group * nd * 24);
        if (rc != IOPX_OK) return rc;
        for (size_t c0 = 0; c0 < cosets; c0 += group) {
            const size_t nc = cosets - c0 < group ? cosets - c0 : group;
            for (size_t i = 0; i < ups.size(); ++i) {
                rc = launch_upper(ups[i], i == 0 ? src : scratch.u64(), scratch.u64(), i == 0 ? 1 : 0, nc, coset_begin + c0);
                if (rc != IOPX_OK) return rc;
            }
            rc = launch_edge(scratch.u64(), dst + 3 * c0 * nd, 0, nc, coset_begin + c0);
            if (rc != IOPX_OK) return rc;
        }
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}


// Forward phase 2 of `batch` (2..4) polynomials over the same cosets: the upper passes run per polynomial into its own scratch, the last
// pass once for all of them (k_bfly_edge_fwd_batch).  Only the shape the prover's extensions have (several cosets, at least one upper
// pass); everything else goes through run_phase2 one polynomial at a time.
static int run_phase2_fwd_batch(AddPlan &pl, const uint64_t *const *srcs, uint64_t *const *dsts, size_t batch, int nhi, size_t coset_begin, size_t coset_count)
{
    const int d = pl.d;
    const size_t cosets = coset_count ? coset_count : ((size_t)1 << nhi);
    const bool batched_edge = tuning().comb && env_int("IOPX_EDGE_BATCH", 1, 0, 1) && batch >= 2 && batch <= 4 && nhi > 0 && d >= EDGE_TILE_BITS + 1 &&
                              cosets * (size_t)d <= ((size_t)1 << tuning().rs_comb_cap_log2);
    if (!batched_edge) {
        for (size_t k = 0; k < batch; ++k) {
            const int rc = run_phase2<false>(pl, srcs[k], dsts[k], nhi, coset_begin, coset_count);
            if (rc != IOPX_OK) return rc;
        }
        return IOPX_OK;
    }
    const size_t nd = (size_t)1 << d;
    size_t group = SCRATCH_BYTES / (nd * 24);
    if (group < 1) group = 1;
    if (group > cosets) group = cosets;
    group = (cosets + (cosets + group - 1) / group - 1) / ((cosets + group - 1) / group);         // even groups (see run_phase2)
    std::vector<std::unique_ptr<TmpBuf>> scratch;
    for (size_t k = 0; k < batch; ++k) {
        scratch.emplace_back(new TmpBuf());
        const int rc = scratch.back()->alloc(group * nd * 24);
        if (rc != IOPX_OK) return rc;
    }
    TmpBuf rs_comb;                                             // combined shift terms of every coset of the call, shared by all launches
    {
        const int rc = rs_comb.alloc(cosets * d * 24);
        if (rc != IOPX_OK) return rc;
        ProfScope ps_("k_rs_combine");
        hipLaunchKernelGGL(k_rs_combine, dim3(grid_for(cosets * d, 256)), dim3(256), 0, stream(), rs_comb.u64(), pl.rs_cur, d, nhi, coset_begin, cosets * (size_t)d);
    }
    for (size_t c0 = 0; c0 < cosets; c0 += group) {
        const size_t nc = cosets - c0 < group ? cosets - c0 : group;
        // upper passes only: the scratch holds the block-order input of the last pass.  (One launch per pass for all polynomials of the batch —
        // four times the workgroups, a quarter of the launch tails — was measured in round 5: k_bfly_upper unchanged, the last pass 0.4 ms slower,
        // its input no longer in the Infinity Cache: profiles/r05_ab_upper_batch.txt.)
        for (size_t k = 0; k < batch; ++k) {
            const int rc = run_phase2<false>(pl, srcs[k], scratch[k]->u64(), nhi, coset_begin + c0, nc, /*upper_only=*/true, rs_comb.u64() + 3 * c0 * d);
            if (rc != IOPX_OK) return rc;
        }
        const P2Geom g = phase2_geom(d);
        BfBatchParams q;
        memset(&q, 0, sizeof(q));
        BfParams &p = q.p;
        p.ltab = pl.ltab.u64(); p.rs = pl.rs_cur; p.d = d; p.nhi = nhi;
        p.a_low = g.a_low;
        set_small_last(p, pl);
        // 512 threads and batch * tile / 2 butterflies per level: whole trips for 2 and 3 tiles of 1024 (48 and 72 KB of LDS, two workgroups
        // per CU either way); four tiles take the half size (48 KB)
        p.c_top = batch == 4 ? g.c_top - 1 : g.c_top;
        p.g_bits = 0;
        const int tb = p.a_low + p.c_top;
        p.total_units = nc << (d - tb);
        p.coset_base = coset_begin + c0;
        int rc;
        p.rs_comb = rs_comb.u64() + 3 * c0 * d;
        q.batch = (int)batch;
        for (size_t k = 0; k < batch; ++k) { q.srcs[k] = scratch[k]->u64(); q.dsts[k] = dsts[k] + 3 * c0 * nd; }
        const size_t lds = (((size_t)24) << tb) * batch;
        if ((rc = set_lds(k_bfly_edge_fwd_batch, lds)) != IOPX_OK) return rc;
        ProfScope ps_("k_bfly_edge_fwd_batch", batch * (nc << d) * 48, batch * ((nc << d) >> 1) * (size_t)p.a_low);
        hipLaunchKernelGGL(k_bfly_edge_fwd_batch, dim3((unsigned)p.total_units), dim3(512), lds, stream(), q);
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

static int check_basis_args(const uint64_t *basis, size_t m, const uint64_t *shift)
{
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    if ((m > 0 && !basis) || !shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null basis/shift");
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_clear_plans(void)
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    (void)hipDeviceSynchronize();
    g_plans.clear();
    clear_mul_plans();
    clear_dist_plans();
    clear_poseidon_sets();
    clear_domain_tables();
    tmp_trim();
    return IOPX_OK;
}

// Cosets [coset_begin, coset_begin + coset_count) of span(basis[0..d)), d = ceil(log2 n_coeffs): the
// contiguous output block [coset_begin * 2^d, (coset_begin + coset_count) * 2^d) of the full transform.
int iopx_add_lde_gf192_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                           const uint64_t *shift, size_t coset_begin, size_t coset_count, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = check_basis_args(basis, m, shift);
    if (rc != IOPX_OK) return rc;
    const size_t n = (size_t)1 << m;
    if (n_coeffs > n) return fail(IOPX_ERR_INVALID_ARGUMENT, "additive FFT: %zu coefficients exceed the domain size %zu", n_coeffs, n);
    if (!d_out || (n_coeffs && !d_coeffs)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    const int d = n_coeffs <= 1 ? 0 : (int)ceil_log2(n_coeffs);
    const int nhi = (int)m - d;
    const size_t all_cosets = (size_t)1 << nhi;
    if (coset_count == 0 || coset_begin >= all_cosets || coset_count > all_cosets - coset_begin)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "coset range [%zu, +%zu) outside the %zu cosets of the transform", coset_begin, coset_count, all_cosets);
    if (n_coeffs <= 1) {
        { ProfScope ps_("k_fill"); hipLaunchKernelGGL(k_fill, dim3(grid_for(coset_count, 256)), dim3(256), 0, stream(), d_out, d_coeffs, (int)(n_coeffs == 1), coset_count); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    }
    AddPlan *pl = nullptr;
    rc = get_plan(basis, d, &pl);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(shift), basis + 3 * (size_t)d, nhi);
    if (rc != IOPX_OK) return rc;

    // phase 1 (and, for a full-size transform, the upper butterfly passes) run in a work buffer: the
    // last pass permutes into natural order and therefore writes out of place
    const size_t nd = (size_t)1 << d;
    TmpBuf work;
    rc = work.alloc(nd * 24);
    if (rc != IOPX_OK) return rc;
    uint64_t *W = work.u64();
    { ProfScope ps_("k_pad_copy"); hipLaunchKernelGGL(k_pad_copy, dim3(grid_for(3 * nd, 256)), dim3(256), 0, stream(), W, d_coeffs, n_coeffs, nd); }
    rc = run_phase1<false>(*pl, W);
    if (rc != IOPX_OK) return rc;
    rc = run_phase2<false>(*pl, W, d_out, nhi, coset_begin, coset_count);
    if (rc != IOPX_OK) return rc;
    return IOPX_OK;                                     // the work buffer is released in stream order
}

int iopx_add_fft_gf192_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                           const uint64_t *shift, uint64_t *d_out)
{
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    const size_t n = (size_t)1 << m;
    if (n_coeffs > n) return fail(IOPX_ERR_INVALID_ARGUMENT, "additive FFT: %zu coefficients exceed the domain size %zu", n_coeffs, n);
    const int d = n_coeffs <= 1 ? 0 : (int)ceil_log2(n_coeffs);
    return iopx_add_lde_gf192_dev(d_coeffs, n_coeffs, basis, m, shift, 0, (size_t)1 << ((int)m - d), d_out);
}

int iopx_add_ifft_gf192_dev(const uint64_t *d_evals, const uint64_t *basis, size_t m, const uint64_t *shift,
                            uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = check_basis_args(basis, m, shift);
    if (rc != IOPX_OK) return rc;
    if (!d_evals || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (m == 0) {
        { const int crc_ = iopx::copy_d2d(d_out, d_evals, 24); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }
    AddPlan *pl = nullptr;
    rc = get_plan(basis, (int)m, &pl);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(shift), nullptr, 0);
    if (rc != IOPX_OK) return rc;
    const uint64_t *src = d_evals;
    TmpBuf tmp;
    if (d_evals == d_out) {     // the first pass permutes: it cannot run in place
        rc = tmp.alloc(((size_t)24) << m);
        if (rc != IOPX_OK) return rc;
        { const int crc_ = iopx::copy_d2d(tmp.p, d_evals, ((size_t)24) << m); if (crc_ != IOPX_OK) return crc_; }
        src = tmp.u64();
    }
    rc = run_phase2<true>(*pl, src, d_out, 0);
    if (rc != IOPX_OK) return rc;
    rc = run_phase1<true>(*pl, d_out);
    if (rc != IOPX_OK) return rc;
    return IOPX_OK;
}

// `batch` inverse transforms over the same domain, vectors stored back to back (d_evals and d_out: batch * 2^m elements)
int iopx_add_ifft_gf192_batch_dev(const uint64_t *d_evals, size_t batch, const uint64_t *basis, size_t m, const uint64_t *shift,
                                  uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = check_basis_args(basis, m, shift);
    if (rc != IOPX_OK) return rc;
    if (!d_evals || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (batch == 0 || batch > 65535) return fail(IOPX_ERR_INVALID_ARGUMENT, "batch size %zu outside 1..65535", batch);
    const size_t bytes = (batch * 24) << m;
    if (m == 0) {
        { const int crc_ = iopx::copy_d2d(d_out, d_evals, bytes); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }
    AddPlan *pl = nullptr;
    rc = get_plan(basis, (int)m, &pl);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(shift), nullptr, 0);
    if (rc != IOPX_OK) return rc;
    const uint64_t *src = d_evals;
    TmpBuf tmp;
    if (d_evals == d_out) {     // the first pass permutes: it cannot run in place
        rc = tmp.alloc(bytes);
        if (rc != IOPX_OK) return rc;
        { const int crc_ = iopx::copy_d2d(tmp.p, d_evals, bytes); if (crc_ != IOPX_OK) return crc_; }
        src = tmp.u64();
    }
    rc = run_phase2<true>(*pl, src, d_out, 0, 0, batch);
    if (rc != IOPX_OK) return rc;
    return run_phase1<true>(*pl, d_out, batch);
}

// `batch` low-degree extensions over the same domain: polynomial k (n_coeffs coefficients at d_coeffs[k]) -> cosets
// [coset_begin, +coset_count) of its transform at d_outs[k].  Phase 1 runs once for the whole batch.
int iopx_add_lde_gf192_batch_dev(const uint64_t *const *d_coeffs, size_t n_coeffs, size_t batch, const uint64_t *basis, size_t m,
                                 const uint64_t *shift, size_t coset_begin, size_t coset_count, uint64_t *const *d_outs)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = check_basis_args(basis, m, shift);
    if (rc != IOPX_OK) return rc;
    if (!d_coeffs || !d_outs) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (batch == 0 || batch > 65535) return fail(IOPX_ERR_INVALID_ARGUMENT, "batch size %zu outside 1..65535", batch);
    for (size_t k = 0; k < batch; ++k) if (!d_coeffs[k] || !d_outs[k]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    const size_t n = (size_t)1 << m;
    if (n_coeffs > n) return fail(IOPX_ERR_INVALID_ARGUMENT, "additive FFT: %zu coefficients exceed the domain size %zu", n_coeffs, n);
    if (n_coeffs <= 1 || batch == 1) {
        for (size_t k = 0; k < batch; ++k) {
            rc = iopx_add_lde_gf192_dev(d_coeffs[k], n_coeffs, basis, m, shift, coset_begin, coset_count, d_outs[k]);
            if (rc != IOPX_OK) return rc;
        }
        return IOPX_OK;
    }
    const int d = (int)ceil_log2(n_coeffs);
    const int nhi = (int)m - d;
    const size_t all_cosets = (size_t)1 << nhi;
    if (coset_count == 0 || coset_begin >= all_cosets || coset_count > all_cosets - coset_begin)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "coset range [%zu, +%zu) outside the %zu cosets of the transform", coset_begin, coset_count, all_cosets);
    AddPlan *pl = nullptr;
    rc = get_plan(basis, d, &pl);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(shift), basis + 3 * (size_t)d, nhi);
    if (rc != IOPX_OK) return rc;
    const size_t nd = (size_t)1 << d;
    TmpBuf work;
    rc = work.alloc(batch * nd * 24);
    if (rc != IOPX_OK) return rc;
    for (size_t k = 0; k < batch; ++k) {
        ProfScope ps_("k_pad_copy");
        hipLaunchKernelGGL(k_pad_copy, dim3(grid_for(3 * nd, 256)), dim3(256), 0, stream(), work.u64() + 3 * k * nd, d_coeffs[k], n_coeffs, nd);
    }
    rc = run_phase1<false>(*pl, work.u64(), batch);
    if (rc != IOPX_OK) return rc;
    for (size_t k0 = 0; k0 < batch; k0 += 4) {               // groups of up to four share the last pass
        const size_t nb = batch - k0 < 4 ? batch - k0 : 4;
        const uint64_t *srcs[4];
        uint64_t *dsts[4];
        for (size_t k = 0; k < nb; ++k) { srcs[k] = work.u64() + 3 * (k0 + k) * nd; dsts[k] = d_outs[k0 + k]; }
        rc = run_phase2_fwd_batch(*pl, srcs, dsts, nb, nhi, coset_begin, coset_count);
        if (rc != IOPX_OK) return rc;
    }
    return IOPX_OK;
}

// FFT_over_field_subset(IFFT_over_field_subset(evals, H), L) for `batch` vectors at once, H = span(basis[0..d)) + eval_shift and
// L = span(basis[0..m)) + shift — how the reference moves a vector of evaluations over a systematic domain onto the codeword
// domain (r1cs_rs_iop.tcc:459-478, basic_lincheck_aux.tcc:94-118, fractal_indexer.tcc:123-156).  Gao-Mateer's phase 1 (twists and
// Taylor expansions) depends on the basis only, not on the shift, so the inverse transform's last half and the forward
// transform's first half cancel exactly: the butterflies of H are undone into block order and the butterflies of every coset of L
// are applied to that — no phase-1 pass at all, same field elements.
int iopx_add_reextend_gf192_batch_dev(const uint64_t *d_evals, size_t batch, const uint64_t *basis, size_t m, size_t d_dim, const uint64_t *eval_shift,
                                      const uint64_t *shift, size_t coset_begin, size_t coset_count, uint64_t *const *d_outs)
{
    return iopx_add_reextend_lde_gf192_batch_dev(d_evals, batch, nullptr, 0, 0, basis, m, d_dim, eval_shift, shift, coset_begin, coset_count, d_outs);
}

// ... together with `n_polys` polynomials given by coefficients (fewer than 2^d_dim + 1 of them): their forward transforms over the same
// cosets join the batch after their own phase 1, so that one set of last passes serves evaluations and polynomials alike
// (r1cs_rs_iop.tcc:567-568 extends f_w right before f_Az, f_Bz, f_Cz).  d_outs holds the `batch` re-extensions, then the n_polys codewords.
int iopx_add_reextend_lde_gf192_batch_dev(const uint64_t *d_evals, size_t batch, const uint64_t *const *d_coeffs, size_t n_coeffs, size_t n_polys,
                                          const uint64_t *basis, size_t m, size_t d_dim, const uint64_t *eval_shift, const uint64_t *shift,
                                          size_t coset_begin, size_t coset_count, uint64_t *const *d_outs)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = check_basis_args(basis, m, shift);
    if (rc != IOPX_OK) return rc;
    if (!d_evals || !d_outs || !eval_shift || (n_polys && !d_coeffs)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (batch == 0 || batch > 65535 || n_polys > 65535) return fail(IOPX_ERR_INVALID_ARGUMENT, "batch size %zu + %zu outside 1..65535", batch, n_polys);
    if (d_dim > m) return fail(IOPX_ERR_INVALID_ARGUMENT, "the evaluation domain must be spanned by the first basis vectors of the codeword domain");
    for (size_t k = 0; k < batch + n_polys; ++k) if (!d_outs[k]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    for (size_t k = 0; k < n_polys; ++k) if (!d_coeffs[k]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    const int d = (int)d_dim, nhi = (int)m - d;
    const size_t all_cosets = (size_t)1 << nhi, nd = (size_t)1 << d;
    if (n_polys && n_coeffs > nd) return fail(IOPX_ERR_INVALID_ARGUMENT, "%zu coefficients exceed the 2^%d of the evaluation domain", n_coeffs, d);
    if (coset_count == 0 || coset_begin >= all_cosets || coset_count > all_cosets - coset_begin)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "coset range [%zu, +%zu) outside the %zu cosets of the transform", coset_begin, coset_count, all_cosets);
    if (d == 0) {       // constants: every evaluation equals them
        for (size_t k = 0; k < batch + n_polys; ++k) {
            rc = iopx_add_lde_gf192_dev(k < batch ? d_evals + 3 * k : d_coeffs[k - batch], k < batch ? 1 : n_coeffs, basis, m, shift, coset_begin, coset_count, d_outs[k]);
            if (rc != IOPX_OK) return rc;
        }
        return IOPX_OK;
    }
    AddPlan *pl = nullptr;
    rc = get_plan(basis, d, &pl);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(eval_shift), nullptr, 0);
    if (rc != IOPX_OK) return rc;
    const size_t total = batch + n_polys;
    TmpBuf work;
    rc = work.alloc(total * nd * 24);
    if (rc != IOPX_OK) return rc;
    rc = run_phase2<true>(*pl, d_evals, work.u64(), 0, 0, batch);                  // H's butterflies undone: natural order -> block order
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(shift), basis + 3 * (size_t)d, nhi);    // stream-ordered: after the inverse passes
    if (rc != IOPX_OK) return rc;
    if (n_polys) {
        for (size_t k = 0; k < n_polys; ++k) {
            ProfScope ps_("k_pad_copy");
            hipLaunchKernelGGL(k_pad_copy, dim3(grid_for(3 * nd, 256)), dim3(256), 0, stream(), work.u64() + 3 * (batch + k) * nd, d_coeffs[k], n_coeffs, nd);
        }
        rc = run_phase1<false>(*pl, work.u64() + 3 * batch * nd, n_polys);
        if (rc != IOPX_OK) return rc;
    }
    for (size_t k0 = 0; k0 < total; k0 += 4) {               // groups of up to four share the last pass
        const size_t nb = total - k0 < 4 ? total - k0 : 4;
        const uint64_t *srcs[4];
        uint64_t *dsts[4];
        for (size_t k = 0; k < nb; ++k) { srcs[k] = work.u64() + 3 * (k0 + k) * nd; dsts[k] = d_outs[k0 + k]; }
        rc = run_phase2_fwd_batch(*pl, srcs, dsts, nb, nhi, coset_begin, coset_count);
        if (rc != IOPX_OK) return rc;
    }
    return IOPX_OK;
}

// Two groups of evaluation vectors over cosets of the same span with different shifts (f_Az, f_Bz, f_Cz over H and f_w over the first coset of
// V inside L), re-extended in ONE batch: each group's butterflies are undone with its own shift terms, the forward passes — and the last pass that up
// to four vectors share — run over all of them.  Outputs: group a's codewords first, then group b's; bit for bit the separate calls'.
int iopx_add_reextend2_gf192_batch_dev(const uint64_t *d_evals_a, size_t batch_a, const uint64_t *eval_shift_a, const uint64_t *d_evals_b, size_t batch_b,
                                       const uint64_t *eval_shift_b, const uint64_t *basis, size_t m, size_t d_dim, const uint64_t *shift,
                                       size_t coset_begin, size_t coset_count, uint64_t *const *d_outs)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = check_basis_args(basis, m, shift);
    if (rc != IOPX_OK) return rc;
    if (!d_evals_a || !d_evals_b || !d_outs || !eval_shift_a || !eval_shift_b) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (batch_a == 0 || batch_b == 0 || batch_a + batch_b > 65535) return fail(IOPX_ERR_INVALID_ARGUMENT, "batch sizes %zu + %zu outside 1..65535", batch_a, batch_b);
    if (d_dim == 0 || d_dim > m) return fail(IOPX_ERR_INVALID_ARGUMENT, "the evaluation domains must be spanned by the first basis vectors of the codeword domain");
    const size_t total = batch_a + batch_b;
    for (size_t k = 0; k < total; ++k) if (!d_outs[k]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    const int d = (int)d_dim, nhi = (int)m - d;
    const size_t all_cosets = (size_t)1 << nhi, nd = (size_t)1 << d;
    if (coset_count == 0 || coset_begin >= all_cosets || coset_count > all_cosets - coset_begin)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "coset range [%zu, +%zu) outside the %zu cosets of the transform", coset_begin, coset_count, all_cosets);
    AddPlan *pl = nullptr;
    rc = get_plan(basis, d, &pl);
    if (rc != IOPX_OK) return rc;
    TmpBuf work;
    rc = work.alloc(total * nd * 24);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(eval_shift_a), nullptr, 0);
    if (rc != IOPX_OK) return rc;
    rc = run_phase2<true>(*pl, d_evals_a, work.u64(), 0, 0, batch_a);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(eval_shift_b), nullptr, 0);            // stream-ordered (or another cached table): after group a's inverse passes
    if (rc != IOPX_OK) return rc;
    rc = run_phase2<true>(*pl, d_evals_b, work.u64() + 3 * batch_a * nd, 0, 0, batch_b);
    if (rc != IOPX_OK) return rc;
    rc = upload_rs(*pl, hgf192::from_words(shift), basis + 3 * (size_t)d, nhi);
    if (rc != IOPX_OK) return rc;
    for (size_t k0 = 0; k0 < total; k0 += 4) {               // groups of up to four share the last pass
        const size_t nb = total - k0 < 4 ? total - k0 : 4;
        const uint64_t *srcs[4];
        uint64_t *dsts[4];
        for (size_t k = 0; k < nb; ++k) { srcs[k] = work.u64() + 3 * (k0 + k) * nd; dsts[k] = d_outs[k0 + k]; }
        rc = run_phase2_fwd_batch(*pl, srcs, dsts, nb, nhi, coset_begin, coset_count);
        if (rc != IOPX_OK) return rc;
    }
    return IOPX_OK;
}

int iopx_add_fft_gf192(const uint64_t *coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                       const uint64_t *shift, uint64_t *out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    const size_t n = (size_t)1 << m;
    if (n_coeffs > n) return fail(IOPX_ERR_INVALID_ARGUMENT, "additive FFT: %zu coefficients exceed the domain size %zu", n_coeffs, n);
    DevBuf din, dout;
    if ((rc = din.alloc(n_coeffs * 24)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(n * 24)) != IOPX_OK) return rc;
    if (n_coeffs) IOPX_HIP(copy_h2d(din.p, coeffs, n_coeffs * 24, stream()));
    rc = iopx_add_fft_gf192_dev(din.u64(), n_coeffs, basis, m, shift, dout.u64());
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(out, dout.p, n * 24, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

int iopx_add_ifft_gf192(const uint64_t *evals, const uint64_t *basis, size_t m, const uint64_t *shift,
                        uint64_t *out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    const size_t n = (size_t)1 << m;
    DevBuf din, dout;
    if ((rc = din.alloc(n * 24)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(n * 24)) != IOPX_OK) return rc;
    IOPX_HIP(copy_h2d(din.p, evals, n * 24, stream()));
    rc = iopx_add_ifft_gf192_dev(din.u64(), basis, m, shift, dout.u64());
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(out, dout.p, n * 24, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

// d_out[i] = d_a[i] * d_b[i]
__global__ void k_gf192_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        gf_store(out, i, gf_mul(gf_load(a, i), gf_load(b, i)));
    }
}

// ---- building blocks of a transform sharded across GPUs (libiop_amd/dist.py) ---------------------------------
// One cross-block butterfly level (fft.tcc:116-117 with stride >= the shard size): tw_i = shift_term + sum_k
// bit_k(index_base + i) * B[k];  lower half: out = a + tw * b;  upper half: out = (a + tw * b) + b.
struct CombineParams {
    const uint64_t *a, *b, *consts;     // consts[0] = shift term, consts[1 + k] = B[k]
    uint64_t *out;
    size_t count, index_base;
    int nb, upper;
};

__global__ void k_combine(CombineParams p)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.count; i += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = p.index_base + i;
        gf192 tw = gf_load(p.consts, 0);
        for (int k = 0; k < p.nb; ++k) {
            if ((idx >> k) & 1) gf_add_to(tw, gf_load(p.consts, 1 + k));
        }
        const gf192 a = gf_load(p.a, i), b = gf_load(p.b, i);
        gf192 r = gf_add(a, gf_mul(b, tw));
        if (p.upper) gf_add_to(r, b);
        gf_store(p.out, i, r);
    }
}

// the level undone (fft.tcc:150-163 with stride >= the shard size): from lo = a + tw * b and up = lo + b,  upper = 0: out = a = lo + tw * (lo + up),
// upper = 1: out = b = lo + up
__global__ void k_combine_inv(CombineParams p)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.count; i += (size_t)gridDim.x * blockDim.x) {
        const gf192 lo = gf_load(p.a, i), up = gf_load(p.b, i);
        const gf192 b = gf_add(lo, up);
        if (p.upper) { gf_store(p.out, i, b); continue; }
        const size_t idx = p.index_base + i;
        gf192 tw = gf_load(p.consts, 0);
        for (int k = 0; k < p.nb; ++k) {
            if ((idx >> k) & 1) gf_add_to(tw, gf_load(p.consts, 1 + k));
        }
        gf_store(p.out, i, gf_add(lo, gf_mul(b, tw)));
    }
}

static int combine_common(bool inverse, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count, size_t index_base,
                          const uint64_t *basis, size_t nb, const uint64_t *shift_term, int upper);

int iopx_add_combine_gf192_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count, size_t index_base,
                               const uint64_t *basis, size_t nb, const uint64_t *shift_term, int upper)
{
    return combine_common(false, d_a, d_b, d_out, count, index_base, basis, nb, shift_term, upper);
}

int iopx_add_combine_inv_gf192_dev(const uint64_t *d_lo, const uint64_t *d_up, uint64_t *d_out, size_t count, size_t index_base,
                                   const uint64_t *basis, size_t nb, const uint64_t *shift_term, int upper)
{
    return combine_common(true, d_lo, d_up, d_out, count, index_base, basis, nb, shift_term, upper);
}

static int combine_common(bool inverse, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count, size_t index_base,
                          const uint64_t *basis, size_t nb, const uint64_t *shift_term, int upper)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_a || !d_b || !d_out || !shift_term || (nb && !basis) || nb > 62) return fail(IOPX_ERR_INVALID_ARGUMENT, "bad argument");
    if (count == 0) return IOPX_OK;
    std::vector<uint64_t> hc(shift_term, shift_term + 3);
    hc.insert(hc.end(), basis, basis + 3 * nb);
    TmpBuf dc;
    if ((rc = dc.alloc(hc.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, hc.data(), hc.size() * 8)) != IOPX_OK) return rc;
    CombineParams p;
    p.a = d_a; p.b = d_b; p.consts = dc.u64(); p.out = d_out; p.count = count; p.index_base = index_base; p.nb = (int)nb; p.upper = upper;
    if (inverse) { ProfScope ps_("k_combine_inv"); hipLaunchKernelGGL(k_combine_inv, dim3(grid_for(count, 256)), dim3(256), 0, stream(), p); }
    else { ProfScope ps_("k_combine"); hipLaunchKernelGGL(k_combine, dim3(grid_for(count, 256)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// In place on 2^log_n elements: S[i] *= d_twist[i] (skipped when d_twist is null), then the Taylor-expansion network of
// one Gao–Mateer level over ALL index bits (ops k = log_n - 2 .. 0, fft.tcc:73-83 with j = 0).
static int taylor_common(bool inverse, uint64_t *d_S, size_t log_n, const uint64_t *d_twist);

int iopx_add_taylor_gf192_dev(uint64_t *d_S, size_t log_n, const uint64_t *d_twist) { return taylor_common(false, d_S, log_n, d_twist); }

// the inverse: the network undone (ops k = 0 .. log_n - 2, fft.tcc:172-188 with j = 0), then S[i] *= d_twist[i] (the inverse powers)
int iopx_add_taylor_inv_gf192_dev(uint64_t *d_S, size_t log_n, const uint64_t *d_twist) { return taylor_common(true, d_S, log_n, d_twist); }

static int taylor_common(bool inverse, uint64_t *d_S, size_t log_n, const uint64_t *d_twist)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_S || log_n > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "bad argument");
    const int d = (int)log_n;
    std::vector<P1Pass> sched;
    if (d <= P1_TILE_BITS) {
        sched.push_back({0, 0, d, 0, 0, d - 2, 0, -1, 0});
    } else {
        for (const P1Pass &ps : phase1_schedule(d)) if (ps.j0 == 0 && ps.j1 == 0) sched.push_back(ps);
    }
    if (inverse) std::reverse(sched.begin(), sched.end());
    for (const P1Pass &ps : sched) {
        P1Params p;
        p.S = d_S; p.pow = d_twist;
        p.d = d; p.c = ps.c; p.h = ps.h; p.A = ps.A;
        p.j0 = 0; p.j1 = 0; p.k_start = ps.k_start; p.k_end = ps.k_end;
        p.xcd_remap = 0;
        p.comb = 0; p.extra = -1; p.skip_j0 = 0;            // the caller's table is per element: nothing is shared or moved
        const int tbits = ps.c + ps.A;
        const size_t lds = ((size_t)24) << tbits;
        const size_t blocks = (size_t)1 << (d - tbits);
        const int threads = (1 << tbits) >= 4 * BLOCK_THREADS ? BLOCK_THREADS : ((1 << tbits) >= 256 ? 64 * ((1 << tbits) / 256) : 64);
        if ((rc = set_lds(k_phase1<false>, lds)) != IOPX_OK || (rc = set_lds(k_phase1<true>, lds)) != IOPX_OK) return rc;
        if (inverse) { ProfScope ps_("k_phase1_inv"); hipLaunchKernelGGL(k_phase1<true>, dim3((unsigned)blocks), dim3(threads), lds, stream(), p); }
        else { ProfScope ps_("k_phase1_fwd"); hipLaunchKernelGGL(k_phase1<false>, dim3((unsigned)blocks), dim3(threads), lds, stream(), p); }
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// d_out[i] = d_a[i] * c, c the same for every element (exercises the wave-uniform comb multiplier)
__global__ void __launch_bounds__(256) k_gf192_mul_uniform(const uint64_t *a, const uint64_t *c, uint64_t *out, size_t count)
{
    const gf192 cc = gf_load(c, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        gf_store(out, i, gf_mul_uniform(gf_load(a, i), cc));
    }
}

int iopx_gf192_mul_uniform_dev(const uint64_t *d_a, const uint64_t *d_c, uint64_t *d_out, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    { ProfScope ps_("k_gf192_mul_uniform"); hipLaunchKernelGGL(k_gf192_mul_uniform, dim3(grid_for(count, 256)), dim3(256), 0, stream(), d_a, d_c, d_out, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// d_out[l] = init * base^l for l < count
int iopx_gf192_pow_table_dev(uint64_t *d_out, size_t count, const uint64_t *base, const uint64_t *init)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_out || !base || !init) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (count == 0) return IOPX_OK;
    const int nb = (int)ceil_log2(count);
    std::vector<uint64_t> sq;
    hgf192 x = hgf192::from_words(base);
    for (int k = 0; k < (nb > 0 ? nb : 1); ++k) { sq.insert(sq.end(), x.w, x.w + 3); x = x.squared(); }
    TmpBuf dsq, full, dinit;
    if ((rc = dsq.alloc(sq.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dsq.p, sq.data(), sq.size() * 8)) != IOPX_OK) return rc;
    if ((rc = full.alloc((((size_t)1) << nb) * 24)) != IOPX_OK) return rc;
    if ((rc = dinit.alloc(24)) != IOPX_OK) return rc;
    if ((rc = upload(dinit.p, init, 24)) != IOPX_OK) return rc;
    rc = build_pow_table(full.u64(), dsq.u64(), nb);
    if (rc != IOPX_OK) return rc;
    { ProfScope ps_("k_gf192_mul_uniform"); hipLaunchKernelGGL(k_gf192_mul_uniform, dim3(grid_for(count, 256)), dim3(256), 0, stream(), (const uint64_t *)full.u64(), (const uint64_t *)dinit.u64(), d_out, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// The half-wavefront product (gf_mul_halves: lanes 0..31 of a wavefront by c[0], lanes 32..63 by c[1]) in a context that depends on EXEC, which the
// asm block narrows and restores: only the lanes with (i & lane_mask) != 0 take the branch that holds the product, a ballot right after the
// product counts the lanes still active, and the lanes outside the branch store their input.  d_active[wave] = that count (popcount of the ballot).
__global__ void __launch_bounds__(256) k_gf192_mul_halves(const uint64_t *a, const uint64_t *c, uint64_t *out, uint32_t *active, size_t count, uint32_t lane_mask)
{
    const gf192 c0 = gf_load(c, 0), c1 = gf_load(c, 1);
    // whole wavefronts per trip (count, the block size and the stride are multiples of 64): element i sits in lane i % 64
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        gf192 v = gf_load(a, i);
        if ((uint32_t)i & lane_mask) {                 // divergent when lane_mask has bits below 6
            v = gf_mul_halves(v, c0, c1, lane);
            const unsigned long long live = __ballot(1);
            if ((live & ((1ull << lane) - 1)) == 0) active[i >> 6] = (uint32_t)__popcll(live);       // the lowest active lane reports
        }
        gf_store(out, i, v);
    }
}

int iopx_gf192_mul_halves_dev(const uint64_t *d_a, const uint64_t *d_c2, uint64_t *d_out, uint32_t *d_active, size_t count, uint32_t lane_mask)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    if (count % 64) return fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_gf192_mul_halves_dev: whole wavefronts only");
    { ProfScope ps_("k_gf192_mul_halves"); hipLaunchKernelGGL(k_gf192_mul_halves, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream(), d_a, d_c2, d_out, d_active, count, lane_mask); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_gf192_mul_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    { ProfScope ps_("k_gf192_mul"); hipLaunchKernelGGL(k_gf192_mul, dim3(grid_for(count, 256)), dim3(256), 0, stream(), d_a, d_b, d_out, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // extern "C"
