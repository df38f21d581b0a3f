// The vector-sized steps of the holographic (Fractal) prover that Aurora's kernels do not already cover, on gfx950:
//
//   iopx_*_div_dev                          batch_inverse / batch_inverse_and_mul (libiop/algebra/utils.tcc:57-118) as an elementwise
//                                           quotient: lagrange_polynomial::evaluations_over_field_subset (algebra/polynomials/
//                                           lagrange_polynomial.tcc:66-136), single_boundary_constraint::evaluated_contents (protocols/
//                                           encoded/common/boundary_constraint.tcc:22-63), rational_linear_combination::evaluated_contents
//                                           (common/rational_linear_combination.tcc:183-209)
//   iopx_domain_offsets_*_dev               point - x over a domain: the denominators of the two quotients above
//   iopx_vanishing_evals_*_dev              c - Z_H(x) over a domain (vanishing_polynomial::evaluations_over_field_subset,
//                                           algebra/polynomials/vanishing_polynomial.tcc:97-137), the Lagrange numerator
//   iopx_rational_combine_*_dev             combined_numerator / combined_denominator::evaluated_contents (rational_linear_combination.tcc:13-108)
//   iopx_rational_sumcheck_constraint_*_dev sumcheck_constraint_oracle::evaluated_contents (protocols/encoded/sumcheck/rational_sumcheck.tcc:58-112)
//
// Montgomery's trick runs per lane over the positions tid, tid + T, tid + 2T, ... (T = lanes of the launch): the running
// products are parked in the OUTPUT buffer on the way up and consumed on the way down (coalesced traffic), and the 256 lanes of a
// workgroup combine their totals in an LDS tree so that the workgroup pays for one field inversion.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "gf192_dev.h"
#include "gf192_host.h"
#include "fp3_dev.h"
#include "fp3_host.h"
#include "runtime.h"

namespace iopx {

static int fo_grid(size_t n, size_t cap = 16384)
{
    size_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    return (int)(g ? g : 1);
}

struct DivParams {
    const uint64_t *num;        // nullable: plain inverses
    const uint64_t *den;
    uint64_t *out;              // distinct from num and den
    const uint64_t *consts;     // fp3: see fo_fp_consts
    size_t n;
};

// The lanes of a workgroup share ONE field inversion: every lane parks its chain's total in an LDS tree (level 0: the 256 lane totals, level l:
// products of pairs, 511 nodes), lane 0 inverts the root, and the inverses walk back down (a node's inverse times its sibling's product is its
// other child's inverse).  A lane then pays 4 products per element plus ~26 for the tree, instead of an exponentiation of its own (~270
// products over F_p): the inversion stops dominating when a lane's chain is short (n <= 2^22).  Written over logical lanes (lane = threadIdx.x,
// + blockDim.x, ...) so that any block size runs the same arithmetic.
// a^-1 in GF(2)[x] / (x^192 + x^7 + x^2 + x + 1) for a != 0 by the binary extended Euclidean algorithm on polynomials (u, v, g1, g2 with
// g1 a = u, g2 a = v mod f; the higher-degree one absorbs a shifted copy of the other until u = 1): at most 2 * 192 shift-and-XOR steps on
// four-word values — for the ONE lane that inverts a workgroup's root, instead of the 191 squarings + 13 products of gf_inv.
__device__ inline gf192 gf_inv_euclid(const gf192 &a)
{
    // scalars, not arrays: a dynamically indexed local array would live in scratch memory
    uint64_t u0 = (uint64_t)a.w[0] | ((uint64_t)a.w[1] << 32), u1 = (uint64_t)a.w[2] | ((uint64_t)a.w[3] << 32), u2 = (uint64_t)a.w[4] | ((uint64_t)a.w[5] << 32), u3 = 0;
    uint64_t v0 = 0x87, v1 = 0, v2 = 0, v3 = 1, g0 = 1, g1 = 0, g2 = 0, g3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0;      // g a = u, h a = v (mod f)
#define IOPX_DEG(x0, x1, x2, x3) ((x3) ? 255 - __builtin_clzll(x3) : ((x2) ? 191 - __builtin_clzll(x2) : ((x1) ? 127 - __builtin_clzll(x1) : ((x0) ? 63 - __builtin_clzll(x0) : -1))))
#define IOPX_XOR_SHIFTED(x0, x1, x2, x3, y0, y1, y2, y3, j)                                                     \
    {                                                                                                           \
        const int bs_ = (j) & 63, ws_ = (j) >> 6;                                                               \
        uint64_t t0_ = y0, t1_ = y1, t2_ = y2, t3_ = y3;                                                        \
        if (bs_) { t3_ = (y3 << bs_) | (y2 >> (64 - bs_)); t2_ = (y2 << bs_) | (y1 >> (64 - bs_)); t1_ = (y1 << bs_) | (y0 >> (64 - bs_)); t0_ = y0 << bs_; } \
        if (ws_ == 0) { x0 ^= t0_; x1 ^= t1_; x2 ^= t2_; x3 ^= t3_; }                                           \
        else if (ws_ == 1) { x1 ^= t0_; x2 ^= t1_; x3 ^= t2_; }                                                 \
        else if (ws_ == 2) { x2 ^= t0_; x3 ^= t1_; }                                                            \
        else { x3 ^= t0_; }                                                                                     \
    }
    int du = IOPX_DEG(u0, u1, u2, u3), dv = 192;
    while (du > 0) {
        int j = du - dv;
        if (j < 0) {
            uint64_t t;
            t = u0; u0 = v0; v0 = t; t = u1; u1 = v1; v1 = t; t = u2; u2 = v2; v2 = t; t = u3; u3 = v3; v3 = t;
            t = g0; g0 = h0; h0 = t; t = g1; g1 = h1; h1 = t; t = g2; g2 = h2; h2 = t; t = g3; g3 = h3; h3 = t;
            const int d = du; du = dv; dv = d;
            j = -j;
        }
        IOPX_XOR_SHIFTED(u0, u1, u2, u3, v0, v1, v2, v3, j)
        IOPX_XOR_SHIFTED(g0, g1, g2, g3, h0, h1, h2, h3, j)
        du = IOPX_DEG(u0, u1, u2, u3);
    }
#undef IOPX_DEG
#undef IOPX_XOR_SHIFTED
    gf192 r;
    r.w[0] = (uint32_t)g0; r.w[1] = (uint32_t)(g0 >> 32); r.w[2] = (uint32_t)g1; r.w[3] = (uint32_t)(g1 >> 32); r.w[4] = (uint32_t)g2; r.w[5] = (uint32_t)(g2 >> 32);
    return r;
}

static constexpr unsigned DIV_LANES = 256, DIV_TREE_NODES = 2 * DIV_LANES - 1;
__device__ __forceinline__ unsigned div_level_base(int l) { return 2 * DIV_LANES - ((2 * DIV_LANES) >> l); }

__global__ void __launch_bounds__(256) k_div_gf192(DivParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint64_t *tree = iopx_smem;
    const size_t T = (size_t)gridDim.x * DIV_LANES;
    gf192 one = gf_zero();
    one.w[0] = 1;
    for (unsigned lane = threadIdx.x; lane < DIV_LANES; lane += blockDim.x) {
        gf192 run = one;
        for (size_t j = (size_t)blockIdx.x * DIV_LANES + lane; j < p.n; j += T) {
            gf192 d = gf_load(p.den, j);
            if (gf_is_zero(d)) d = one;
            run = gf_mul(run, d);
            gf_store(p.out, j, run);
        }
        gf_store(tree, lane, run);
    }
    __syncthreads();
    for (int l = 1; (DIV_LANES >> l) >= 1; ++l) {
        const unsigned cnt = DIV_LANES >> l, base = div_level_base(l), below = div_level_base(l - 1);
        for (unsigned i = threadIdx.x; i < cnt; i += blockDim.x) gf_store(tree, base + i, gf_mul(gf_load(tree, below + 2 * i), gf_load(tree, below + 2 * i + 1)));
        __syncthreads();
    }
    if (threadIdx.x == 0) gf_store(tree, DIV_TREE_NODES - 1, gf_inv_euclid(gf_load(tree, DIV_TREE_NODES - 1)));
    __syncthreads();
    for (int l = 8; l >= 1; --l) {
        const unsigned cnt = DIV_LANES >> l, base = div_level_base(l), below = div_level_base(l - 1);
        for (unsigned i = threadIdx.x; i < cnt; i += blockDim.x) {
            const gf192 inv = gf_load(tree, base + i), left = gf_load(tree, below + 2 * i), right = gf_load(tree, below + 2 * i + 1);
            gf_store(tree, below + 2 * i, gf_mul(inv, right));
            gf_store(tree, below + 2 * i + 1, gf_mul(inv, left));
        }
        __syncthreads();
    }
    for (unsigned lane = threadIdx.x; lane < DIV_LANES; lane += blockDim.x) {
        const size_t tid = (size_t)blockIdx.x * DIV_LANES + lane;
        const size_t count = tid < p.n ? (p.n - tid + T - 1) / T : 0;
        gf192 inv = gf_load(tree, lane);
        for (size_t i = count; i-- > 0; ) {
            const size_t j = tid + i * T;
            gf192 d = gf_load(p.den, j);
            const bool zero = gf_is_zero(d);
            if (zero) d = one;
            const gf192 before = i ? gf_load(p.out, j - T) : one;
            gf192 q = gf_mul(before, inv);                      // 1 / d
            inv = gf_mul(inv, d);
            if (p.num) q = gf_mul(q, gf_load(p.num, j));
            gf_store(p.out, j, zero ? gf_zero() : q);           // a zero denominator yields zero (utils.tcc:79-97)
        }
    }
}

// v^-1 mod p for 0 < v < p (plain integers, not Montgomery forms) by the binary extended Euclidean algorithm (HAC 14.61): at most ~2 * 181
// halve-or-subtract steps on 192-bit integers — some 10^4 instructions against the 10^5 of the Fermat power (270 products) — for the ONE lane
// of a workgroup that inverts the root of the tree; data-dependent control flow costs nothing with one lane active.
__device__ inline fp3 fp_inv_binary(const fp3 &v)
{
    const uint64_t P0 = (uint64_t)FP3_P[0] | ((uint64_t)FP3_P[1] << 32), P1 = (uint64_t)FP3_P[2] | ((uint64_t)FP3_P[3] << 32), P2 = (uint64_t)FP3_P[4] | ((uint64_t)FP3_P[5] << 32);
    uint64_t u0 = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32), u1 = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32), u2 = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
    uint64_t w0 = P0, w1 = P1, w2 = P2, a0 = 1, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;      // a u0 = u (mod p), b u0 = w (mod p) for the initial u0
#define IOPX_HALVE(x0, x1, x2) { x0 = (x0 >> 1) | (x1 << 63); x1 = (x1 >> 1) | (x2 << 63); x2 >>= 1; }
#define IOPX_ADD(x0, x1, x2, y0, y1, y2) { const uint64_t s0 = x0 + y0, c0 = s0 < x0, t1 = x1 + y1, c1 = t1 < x1, s1 = t1 + c0, c2 = s1 < t1; x0 = s0; x1 = s1; x2 = x2 + y2 + (c1 | c2); }
#define IOPX_SUB(x0, x1, x2, y0, y1, y2) { const uint64_t d0 = x0 - y0, r0 = x0 < y0, t1 = x1 - y1, r1 = x1 < y1, d1 = t1 - r0, r2 = t1 < r0; x0 = d0; x1 = d1; x2 = x2 - y2 - (r1 | r2); }
#define IOPX_GEQ(x0, x1, x2, y0, y1, y2) (x2 != y2 ? x2 > y2 : (x1 != y1 ? x1 > y1 : x0 >= y0))
    while (!((u0 == 1 && (u1 | u2) == 0) || (w0 == 1 && (w1 | w2) == 0))) {
        while (!(u0 & 1)) {
            IOPX_HALVE(u0, u1, u2)
            if (a0 & 1) IOPX_ADD(a0, a1, a2, P0, P1, P2)
            IOPX_HALVE(a0, a1, a2)
        }
        while (!(w0 & 1)) {
            IOPX_HALVE(w0, w1, w2)
            if (b0 & 1) IOPX_ADD(b0, b1, b2, P0, P1, P2)
            IOPX_HALVE(b0, b1, b2)
        }
        if (IOPX_GEQ(u0, u1, u2, w0, w1, w2)) {
            IOPX_SUB(u0, u1, u2, w0, w1, w2)
            if (!IOPX_GEQ(a0, a1, a2, b0, b1, b2)) IOPX_ADD(a0, a1, a2, P0, P1, P2)
            IOPX_SUB(a0, a1, a2, b0, b1, b2)
        } else {
            IOPX_SUB(w0, w1, w2, u0, u1, u2)
            if (!IOPX_GEQ(b0, b1, b2, a0, a1, a2)) IOPX_ADD(b0, b1, b2, P0, P1, P2)
            IOPX_SUB(b0, b1, b2, a0, a1, a2)
        }
    }
#undef IOPX_HALVE
#undef IOPX_ADD
#undef IOPX_SUB
#undef IOPX_GEQ
    const bool first = u0 == 1 && (u1 | u2) == 0;
    const uint64_t r0 = first ? a0 : b0, r1 = first ? a1 : b1, r2 = first ? a2 : b2;
    fp3 r;
    r.w[0] = (uint32_t)r0; r.w[1] = (uint32_t)(r0 >> 32); r.w[2] = (uint32_t)r1; r.w[3] = (uint32_t)(r1 >> 32); r.w[4] = (uint32_t)r2; r.w[5] = (uint32_t)(r2 >> 32);
    return r;
}

// F_p.  The device product is a b 2^-203 and stored values carry 2^192 (fp3_dev.h), so each product with a stored denominator
// multiplies the running value by d 2^-11: starting from the 2^203 form of 1, run_i = 2^203 (prod_{j<=i} d_j) 2^(-11 i).  The lane totals,
// their tree products and the inverses are all in the 2^203 form, which the device product keeps closed (x^(p-2) included); a lane's inverse
// times 2^-11 once makes before * inv = 2^203 / d_i exactly, at every i — the powers of 2^11 cancel step by step — so the denominators are never
// converted: 4 products per element plus the shared inversion.
__global__ void __launch_bounds__(256) k_div_fp3(DivParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint64_t *tree = iopx_smem;
    const size_t T = (size_t)gridDim.x * DIV_LANES;
    const fp3 k192 = fp_load(p.consts, 1), one_t = fp_load(p.consts, 3), unscale_t = fp_load(p.consts, 4);
    for (unsigned lane = threadIdx.x; lane < DIV_LANES; lane += blockDim.x) {
        fp3 run = one_t;
        for (size_t j = (size_t)blockIdx.x * DIV_LANES + lane; j < p.n; j += T) {
            const fp3 d = fp_load(p.den, j);
            const bool zero = (d.w[0] | d.w[1] | d.w[2] | d.w[3] | d.w[4] | d.w[5]) == 0;
            run = fp_mul(run, zero ? k192 : d);                 // a zero denominator is stepped over as a stored 1
            fp_store(p.out, j, run);
        }
        fp_store(tree, lane, run);
    }
    __syncthreads();
    for (int l = 1; (DIV_LANES >> l) >= 1; ++l) {
        const unsigned cnt = DIV_LANES >> l, base = div_level_base(l), below = div_level_base(l - 1);
        for (unsigned i = threadIdx.x; i < cnt; i += blockDim.x) fp_store(tree, base + i, fp_mul(fp_load(tree, below + 2 * i), fp_load(tree, below + 2 * i + 1)));
        __syncthreads();
    }
    // the root is r 2^203 as an integer; its integer inverse times 2^609 through the device product is 2^203 / r: the 2^203 form of 1 / r
    if (threadIdx.x == 0) fp_store(tree, DIV_TREE_NODES - 1, fp_mul(fp_inv_binary(fp_load(tree, DIV_TREE_NODES - 1)), fp_load(p.consts, 5)));
    __syncthreads();
    for (int l = 8; l >= 1; --l) {
        const unsigned cnt = DIV_LANES >> l, base = div_level_base(l), below = div_level_base(l - 1);
        for (unsigned i = threadIdx.x; i < cnt; i += blockDim.x) {
            const fp3 inv = fp_load(tree, base + i), left = fp_load(tree, below + 2 * i), right = fp_load(tree, below + 2 * i + 1);
            fp_store(tree, below + 2 * i, fp_mul(inv, right));
            fp_store(tree, below + 2 * i + 1, fp_mul(inv, left));
        }
        __syncthreads();
    }
    for (unsigned lane = threadIdx.x; lane < DIV_LANES; lane += blockDim.x) {
        const size_t tid = (size_t)blockIdx.x * DIV_LANES + lane;
        const size_t count = tid < p.n ? (p.n - tid + T - 1) / T : 0;
        fp3 inv = fp_mul(fp_load(tree, lane), unscale_t);
        for (size_t i = count; i-- > 0; ) {
            const size_t j = tid + i * T;
            const fp3 d = fp_load(p.den, j);
            const bool zero = (d.w[0] | d.w[1] | d.w[2] | d.w[3] | d.w[4] | d.w[5]) == 0;
            const fp3 before = i ? fp_load(p.out, j - T) : one_t;
            const fp3 q = fp_mul(before, inv);                  // (1 / d) 2^203
            inv = fp_mul(inv, zero ? k192 : d);
            fp_store(p.out, j, zero ? fp_zero() : fp_mul(p.num ? fp_load(p.num, j) : k192, q));
        }
    }
}

// ---- subset sums over an affine subspace: out[j] = tab[0] + sum_{bit k of j} tab[1 + k] --------------------------------------
__device__ __forceinline__ gf192 fo_subset_sum(const uint64_t *t, int m, uint32_t jlo, uint32_t jhi_uniform)
{
    return subset_sum_ext(t, m, jlo, jhi_uniform);
}

__global__ void __launch_bounds__(256) k_subset_sums_gf192(uint64_t *out, const uint64_t *tab, int m, size_t n)
{
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
        const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 8));
        const size_t end = base + 256 < n ? base + 256 : n;
        for (size_t j = base + threadIdx.x; j < end; j += blockDim.x) gf_store(out, j, fo_subset_sum(tab, m, (uint32_t)(j & 255), jhi));
    }
}

// ---- c - init base^j over a multiplicative coset (two-level table, hi pre-divided by 2^11 so the product is in libff's form) ----
__global__ void __launch_bounds__(256) k_geometric_offsets_fp3(uint64_t *out, const uint64_t *hi, const uint64_t *lo, const uint64_t *c, size_t n)
{
    const fp3 cst = fp_load(c, 0);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        fp_store(out, j, fp_sub(cst, fp_mul(fp_load(hi, j >> 12), fp_load(lo, j & 4095))));
}

// ---- rational linear combination: N = sum_i c_i N_i prod_{k != i} D_k, D = prod_k D_k ----------------------------------------
#define RATIONAL_MAX 4
struct RationalParams {
    const uint64_t *N[RATIONAL_MAX], *D[RATIONAL_MAX];
    const uint64_t *c;          // num coefficients (fp3: 2^203 form), then (fp3) the raw 2^214
    uint64_t *outN, *outD;
    int num;
    size_t n;
};

__global__ void __launch_bounds__(256) k_rational_combine_gf192(RationalParams p)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        gf192 d[RATIONAL_MAX];
        for (int k = 0; k < p.num; ++k) d[k] = gf_load(p.D[k], j);
        gf192 numer = gf_zero(), denom = d[0];
        for (int k = 1; k < p.num; ++k) denom = gf_mul(denom, d[k]);
        for (int i = 0; i < p.num; ++i) {
            gf192 cur = gf_mul_uniform(gf_load(p.N[i], j), gf_load(p.c, i));
            for (int k = 0; k < p.num; ++k) if (k != i) cur = gf_mul(cur, d[k]);
            gf_add_to(numer, cur);
        }
        gf_store(p.outN, j, numer);
        gf_store(p.outD, j, denom);
    }
}

__global__ void __launch_bounds__(256) k_rational_combine_fp3(RationalParams p)
{
    const fp3 k214 = fp_load(p.c, p.num);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        fp3 t[RATIONAL_MAX];                                // the denominators in the 2^203 form
        for (int k = 0; k < p.num; ++k) t[k] = fp_mul(fp_load(p.D[k], j), k214);
        fp3 denom = fp_load(p.D[0], j);
        for (int k = 1; k < p.num; ++k) denom = fp_mul(denom, t[k]);
        fp7w w;                                               // the last factor of each term is accumulated: one reduction for the sum
        fp7w_zero(w);
        for (int i = 0; i < p.num; ++i) {
            fp3 cur = fp_mul(fp_load(p.N[i], j), fp_load(p.c, i));
            int last = -1;
            for (int k = 0; k < p.num; ++k) if (k != i) last = k;
            for (int k = 0; k < p.num; ++k) if (k != i && k != last) cur = fp_mul(cur, t[k]);
            if (last >= 0) fp_mac(w, cur, t[last]);
            else fp_mac(w, cur, fp_load(p.c, p.num + 1));     // a single rational: times the 2^203 form of 1
        }
        const fp3 numer = fp_redc(w);
        fp_store(p.outN, j, numer);
        fp_store(p.outD, j, denom);
    }
}

// ---- rational sumcheck constraint oracle ------------------------------------------------------------------------------------
// affine subspaces: q(x) = (D(x) (p(x) + eps^-1 mu x^(|K| - 1)) - N(x)) / Z_K(x); x^(|K| - 1) = x^|K| / x with x^|K| a subset sum
// and 1 / x supplied by the caller (iopx_gf192_div_dev over iopx_domain_offsets_gf192_dev); Z_K is constant on the cosets of K,
// which are contiguous blocks of the codeword domain (K is spanned by a prefix of its basis), so 1 / Z_K is a small table.
struct ConstraintAddParams {
    const uint64_t *p, *N, *D, *xinv;
    const uint64_t *ktab;       // (m + 1)-entry subset-sum table of x^|K|
    const uint64_t *zinv;       // 2^(m - k) entries
    const uint64_t *c;          // eps^-1 mu
    uint64_t *out;
    int m, k;
    size_t n;
};

__global__ void __launch_bounds__(256) k_sumcheck_constraint_gf192(ConstraintAddParams a)
{
    const gf192 cst = gf_load(a.c, 0);
    for (size_t base = (size_t)blockIdx.x * 256; base < a.n; base += (size_t)gridDim.x * 256) {
        const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 8));
        const size_t end = base + 256 < a.n ? base + 256 : a.n;
        for (size_t j = base + threadIdx.x; j < end; j += blockDim.x) {
            const gf192 xk = fo_subset_sum(a.ktab, a.m, (uint32_t)(j & 255), jhi);
            gf192 s = gf_mul(gf_mul_uniform(xk, cst), gf_load(a.xinv, j));
            gf_add_to(s, gf_load(a.p, j));
            gf192 t = gf_mul(gf_load(a.D, j), s);
            gf_add_to(t, gf_load(a.N, j));
            gf_store(a.out, j, a.k >= 6 ? gf_mul_uniform(t, gf_load(a.zinv, j >> a.k)) : gf_mul(t, gf_load(a.zinv, j >> a.k)));   // one coset per wavefront
        }
    }
}

// multiplicative cosets: q(x) = (D(x) (x p(x) + mu / |K|) - N(x)) / Z_K(x); position j lies in coset j mod (|L| / |K|).
// D (x p + c) is a data x data product (scale 2^181): N is brought to the same scale by the stored 1 and the inverse table
// carries 2^214 (as in k_rowcheck_fp).
__global__ void __launch_bounds__(256) k_sumcheck_constraint_fp3(uint64_t *out, const uint64_t *pp, const uint64_t *N, const uint64_t *D,
                                                                 const uint64_t *xhi, const uint64_t *xlo, const uint64_t *zinv_scaled,
                                                                 const uint64_t *consts, size_t num_cosets, size_t n)
{
    const fp3 c = fp_load(consts, 0), one = fp_load(consts, 1);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const fp3 x = fp_mul(fp_load(xhi, j >> 12), fp_load(xlo, j & 4095));         // 2^203 form
        const fp3 s = fp_add(fp_mul(fp_load(pp, j), x), c);
        fp7w w;                                               // D s - N 1 = D s + (p - N) 1 with one reduction
        fp7w_zero(w);
        fp_mac(w, fp_load(D, j), s);
        fp_mac(w, fp_neg(fp_load(N, j)), one);
        fp_store(out, j, fp_mul(fp_redc(w), fp_load(zinv_scaled, j & (num_cosets - 1))));
    }
}

// slots of three words: 0: 2^214 (raw), 1: 2^192 = the stored 1, 2: p - 2, 3: 2^203 = the device form of 1, 4: the device form of 2^-11,
// 5: 2^609 (raw)
static void fo_fp_consts(uint64_t (&c)[18])
{
    const hfp3 k192 = hfp3::one(), k203 = k192.table_form(), k214 = k203.table_form();
    const hfp3 unscale_t = hfp3::from_uint(2048).inverse().table_form();
    memcpy(c, k214.w, 24); memcpy(c + 3, k192.w, 24);
    c[6] = hfp3::P[0] - 2; c[7] = hfp3::P[1]; c[8] = hfp3::P[2];
    memcpy(c + 9, k203.w, 24);
    memcpy(c + 12, unscale_t.w, 24);
    const hfp3 k609 = hfp3::from_uint(2).pow(417);              // stored words of 2^417: 2^417 2^192
    memcpy(c + 15, k609.w, 24);
}

static int div_common(const uint64_t *d_num, const uint64_t *d_den, uint64_t *d_out, size_t n, bool prime_field)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (n == 0) return IOPX_OK;
    if (!d_den || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (d_out == d_den || d_out == d_num) return fail(IOPX_ERR_INVALID_ARGUMENT, "the quotient buffer holds the running products: it cannot alias an input");
    DivParams p;
    p.num = d_num; p.den = d_den; p.out = d_out; p.consts = nullptr; p.n = n;
    TmpBuf dc;
    if (prime_field) {
        // the field's own constants: computed once per process, kept on the device
        rc = cached_domain_table({ 0x646976 /* "div" */ }, [](std::vector<uint64_t> &w) -> int { uint64_t c[18]; fo_fp_consts(c); w.assign(c, c + 18); return IOPX_OK; }, dc);
        if (rc != IOPX_OK) return rc;
        p.consts = dc.u64();
    }
    // 16 elements per lane while that fills the GPU (the shared inversion is one lane's serial work per workgroup), 2048 workgroups at most
    size_t grid = (n + DIV_LANES * 16 - 1) / (DIV_LANES * 16);
    if (grid > 2048) grid = 2048;
    const size_t bytes = n * 24 * (d_num ? 5 : 4), lds = (size_t)DIV_TREE_NODES * 24 + 8;
    if (prime_field) { ProfScope ps_("k_div_fp3", bytes); hipLaunchKernelGGL(k_div_fp3, dim3((unsigned)grid), dim3(256), lds, stream(), p); }
    else { ProfScope ps_("k_div_gf192", bytes); hipLaunchKernelGGL(k_div_gf192, dim3((unsigned)grid), dim3(256), lds, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

static int subset_sums(const std::vector<uint64_t> &plain, size_t m, uint64_t *d_out)
{
    int rc;
    std::vector<hgf192> entries;
    for (size_t k = 0; k <= m; ++k) entries.push_back(hgf192::from_words(&plain[3 * k]));
    std::vector<uint64_t> tab;
    append_subset_table_with_ext(tab, entries);
    TmpBuf dt;
    if ((rc = dt.alloc(tab.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dt.p, tab.data(), tab.size() * 8)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << m;
    { ProfScope ps_("k_subset_sums_gf192", n * 24); hipLaunchKernelGGL(k_subset_sums_gf192, dim3(fo_grid(n)), dim3(256), 0, stream(), d_out, (const uint64_t *)dt.u64(), (int)m, n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

static int geometric_offsets(const hfp3 &base, const hfp3 &init, const hfp3 &c, size_t log_n, uint64_t *d_out)
{
    int rc;
    TmpBuf hi, lo, dc;
    const hfp3 unscale = hfp3::from_uint(2048).inverse();     // hi * lo comes out as value 2^203; libff's form is value 2^192
    if ((rc = build_two_level(base, init * unscale, (int)log_n, hi, lo)) != IOPX_OK) return rc;
    if ((rc = dc.alloc(24)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, c.w, 24)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << log_n;
    { ProfScope ps_("k_geometric_offsets_fp3", n * 24); hipLaunchKernelGGL(k_geometric_offsets_fp3, dim3(fo_grid(n)), dim3(256), 0, stream(), d_out,
                                                                          (const uint64_t *)hi.u64(), (const uint64_t *)lo.u64(), (const uint64_t *)dc.u64(), n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

static int rational_common(const void *const *d_N, const void *const *d_D, size_t num, const uint64_t *coeffs, size_t n, uint64_t *d_outN,
                           uint64_t *d_outD, bool prime_field)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_N || !d_D || !coeffs || !d_outN || !d_outD) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (num == 0 || num > RATIONAL_MAX) return fail(IOPX_ERR_INVALID_ARGUMENT, "Expected same number of evaluations as in registration.");
    std::vector<uint64_t> hc(coeffs, coeffs + 3 * num);
    if (prime_field) {
        for (size_t i = 0; i < num; ++i) { const hfp3 t = hfp3::from_words(coeffs + 3 * i).table_form(); memcpy(&hc[3 * i], t.w, 24); }
        const hfp3 k203 = hfp3::one().table_form(), k214 = k203.table_form();
        hc.insert(hc.end(), k214.w, k214.w + 3);
        hc.insert(hc.end(), k203.w, k203.w + 3);
    }
    TmpBuf dc;
    if ((rc = dc.alloc(hc.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, hc.data(), hc.size() * 8)) != IOPX_OK) return rc;
    RationalParams p;
    memset(&p, 0, sizeof(p));
    for (size_t i = 0; i < num; ++i) {
        if (!d_N[i] || !d_D[i]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null oracle");
        p.N[i] = (const uint64_t *)d_N[i]; p.D[i] = (const uint64_t *)d_D[i];
    }
    p.c = dc.u64(); p.outN = d_outN; p.outD = d_outD; p.num = (int)num; p.n = n;
    const size_t bytes = n * 24 * (2 * num + 2);
    if (prime_field) { ProfScope ps_("k_rational_combine_fp3", bytes); hipLaunchKernelGGL(k_rational_combine_fp3, dim3(fo_grid(n)), dim3(256), 0, stream(), p); }
    else { ProfScope ps_("k_rational_combine_gf192", bytes); hipLaunchKernelGGL(k_rational_combine_gf192, dim3(fo_grid(n)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_gf192_div_dev(const uint64_t *d_num, const uint64_t *d_den, uint64_t *d_out, size_t count) { return div_common(d_num, d_den, d_out, count, false); }
int iopx_fp3_div_dev(const uint64_t *d_num, const uint64_t *d_den, uint64_t *d_out, size_t count) { return div_common(d_num, d_den, d_out, count, true); }

int iopx_domain_offsets_gf192_dev(const uint64_t *basis, size_t m, const uint64_t *shift, const uint64_t *point, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if ((m > 0 && !basis) || !shift || !point || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    std::vector<uint64_t> tab(3 * (m + 1));
    for (int w = 0; w < 3; ++w) tab[w] = shift[w] ^ point[w];               // point - x = point + shift + sum of basis vectors
    if (m) memcpy(&tab[3], basis, 24 * m);
    return subset_sums(tab, m, d_out);
}

int iopx_domain_offsets_fp3_dev(size_t log_n, const uint64_t *gen, const uint64_t *shift, const uint64_t *point, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!gen || !shift || !point || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    return geometric_offsets(hfp3::from_words(gen), hfp3::from_words(shift), hfp3::from_words(point), log_n, d_out);
}

int iopx_vanishing_evals_gf192_dev(const uint64_t *basis, size_t m, const uint64_t *shift, const uint64_t *vanishing_basis, size_t vanishing_dim,
                                   const uint64_t *vanishing_shift, const uint64_t *constant, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if ((m > 0 && !basis) || !shift || (vanishing_dim > 0 && !vanishing_basis) || !vanishing_shift || !constant || !d_out)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (m > 40 || vanishing_dim > 63) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    const SubspacePoly lin(vanishing_basis, vanishing_dim);
    std::vector<uint64_t> tab(3 * (m + 1));
    const hgf192 t0 = hgf192::from_words(constant) + lin.eval(hgf192::from_words(shift)) + lin.eval(hgf192::from_words(vanishing_shift));
    memcpy(&tab[0], t0.w, 24);
    for (size_t k = 0; k < m; ++k) { const hgf192 t = lin.eval(hgf192::from_words(basis + 3 * k)); memcpy(&tab[3 * (k + 1)], t.w, 24); }
    return subset_sums(tab, m, d_out);
}

int iopx_vanishing_evals_fp3_dev(size_t log_n, const uint64_t *gen, const uint64_t *shift, size_t vanishing_log_order, const uint64_t *vanishing_shift,
                                 const uint64_t *constant, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!gen || !shift || !vanishing_shift || !constant || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (log_n > 31 || vanishing_log_order > 62) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    const uint64_t order = (uint64_t)1 << vanishing_log_order;
    // c - Z_H(x) = (c + shift_H^|H|) - x^|H|
    const hfp3 vp_shift = hfp3::from_words(vanishing_shift).pow(order);
    const hfp3 zero = hfp3();
    const hfp3 c = hfp3::from_words(constant) - (zero - vp_shift);
    return geometric_offsets(hfp3::from_words(gen).pow(order), hfp3::from_words(shift).pow(order), c, log_n, d_out);
}

int iopx_rational_combine_gf192_dev(const void *const *d_numerators, const void *const *d_denominators, size_t num_rationals, const uint64_t *coefficients,
                                    size_t n, uint64_t *d_numerator_out, uint64_t *d_denominator_out)
{
    return rational_common(d_numerators, d_denominators, num_rationals, coefficients, n, d_numerator_out, d_denominator_out, false);
}
int iopx_rational_combine_fp3_dev(const void *const *d_numerators, const void *const *d_denominators, size_t num_rationals, const uint64_t *coefficients,
                                  size_t n, uint64_t *d_numerator_out, uint64_t *d_denominator_out)
{
    return rational_common(d_numerators, d_denominators, num_rationals, coefficients, n, d_numerator_out, d_denominator_out, true);
}

int iopx_rational_sumcheck_constraint_gf192_dev(const uint64_t *d_p, const uint64_t *d_N, const uint64_t *d_D, const uint64_t *d_xinv, const uint64_t *basis,
                                                size_t m, const uint64_t *shift, size_t summation_dim, const uint64_t *summation_shift,
                                                const uint64_t *claimed_sum, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_p || !d_N || !d_D || !d_xinv || !d_out || (m > 0 && !basis) || !shift || !summation_shift || !claimed_sum)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (summation_dim > m || m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "the summation domain must be spanned by a prefix of the codeword domain's basis");
    const size_t k = summation_dim, cosets = (size_t)1 << (m - k);
    const SubspacePoly lin(basis, k);
    if (lin.coeff[0].is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "the summation domain's basis is linearly dependent");
    const hgf192 c = lin.coeff[0].inverse() * hgf192::from_words(claimed_sum);          // eps^-1 mu (rational_sumcheck.tcc:52-56)
    const hgf192 z_shift = lin.eval(hgf192::from_words(summation_shift));
    std::vector<uint64_t> ktab, zinv(3 * cosets);
    std::vector<hgf192> kentries;
    for (size_t i = 0; i <= m; ++i) {
        hgf192 v = hgf192::from_words(i == 0 ? shift : basis + 3 * (i - 1));
        for (size_t s = 0; s < k; ++s) v = v.squared();                                // v^|K|
        kentries.push_back(v);
    }
    append_subset_table_with_ext(ktab, kentries);
    for (size_t cidx = 0; cidx < cosets; ++cidx) {
        hgf192 x = hgf192::from_words(shift);
        for (size_t b = k; b < m; ++b) if ((cidx >> (b - k)) & 1) x += hgf192::from_words(basis + 3 * b);
        const hgf192 z = lin.eval(x) + z_shift;
        if (z.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "the codeword domain intersects the summation domain");
        const hgf192 zi = z.inverse();
        memcpy(&zinv[3 * cidx], zi.w, 24);
    }
    TmpBuf dk, dz, dc;
    if ((rc = dk.alloc(ktab.size() * 8)) != IOPX_OK) return rc;
    if ((rc = dz.alloc(zinv.size() * 8)) != IOPX_OK) return rc;
    if ((rc = dc.alloc(24)) != IOPX_OK) return rc;
    if ((rc = upload(dk.p, ktab.data(), ktab.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dz.p, zinv.data(), zinv.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, c.w, 24)) != IOPX_OK) return rc;
    ConstraintAddParams a;
    a.p = d_p; a.N = d_N; a.D = d_D; a.xinv = d_xinv; a.ktab = dk.u64(); a.zinv = dz.u64(); a.c = dc.u64(); a.out = d_out;
    a.m = (int)m; a.k = (int)k; a.n = (size_t)1 << m;
    { ProfScope ps_("k_sumcheck_constraint_gf192", a.n * 24 * 5); hipLaunchKernelGGL(k_sumcheck_constraint_gf192, dim3(fo_grid(a.n)), dim3(256), 0, stream(), a); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_rational_sumcheck_constraint_fp3_dev(const uint64_t *d_p, const uint64_t *d_N, const uint64_t *d_D, size_t log_n, const uint64_t *gen,
                                              const uint64_t *shift, size_t summation_log_order, const uint64_t *summation_shift,
                                              const uint64_t *claimed_sum, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_p || !d_N || !d_D || !d_out || !gen || !shift || !summation_shift || !claimed_sum) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (summation_log_order > log_n || log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "the summation domain must be a sub-domain of the codeword domain");
    const uint64_t order_k = (uint64_t)1 << summation_log_order;
    const size_t cosets = (size_t)1 << (log_n - summation_log_order);
    const hfp3 g = hfp3::from_words(gen), s = hfp3::from_words(shift), ks = hfp3::from_words(summation_shift);
    const hfp3 vp_shift = ks.pow(order_k), g_k = g.pow(order_k);
    hfp3 cur = s.pow(order_k);
    std::vector<uint64_t> zinv(3 * cosets);
    for (size_t j = 0; j < cosets; ++j) {
        const hfp3 z = cur - vp_shift;
        if (z.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "the codeword domain intersects the summation domain");
        const hfp3 zi = z.inverse().table_form().table_form();
        memcpy(&zinv[3 * j], zi.w, 24);
        cur = cur * g_k;
    }
    const hfp3 c = hfp3::from_uint(order_k).inverse() * hfp3::from_words(claimed_sum);     // mu / |K| (rational_sumcheck.tcc:47-51)
    const hfp3 one = hfp3::one();
    uint64_t consts[6];
    memcpy(consts, c.w, 24); memcpy(consts + 3, one.w, 24);
    TmpBuf xhi, xlo, dz, dc;
    if ((rc = build_two_level(g, s, (int)log_n, xhi, xlo)) != IOPX_OK) return rc;
    if ((rc = dz.alloc(zinv.size() * 8)) != IOPX_OK) return rc;
    if ((rc = dc.alloc(sizeof(consts))) != IOPX_OK) return rc;
    if ((rc = upload(dz.p, zinv.data(), zinv.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, consts, sizeof(consts))) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << log_n;
    { ProfScope ps_("k_sumcheck_constraint_fp3", n * 24 * 4); hipLaunchKernelGGL(k_sumcheck_constraint_fp3, dim3(fo_grid(n)), dim3(256), 0, stream(), d_out, d_p, d_N, d_D,
                                                                             (const uint64_t *)xhi.u64(), (const uint64_t *)xlo.u64(), (const uint64_t *)dz.u64(),
                                                                             (const uint64_t *)dc.u64(), cosets, n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// ---- host helpers: O(dim) field operations for the handful of scalars the protocol layer needs ----
// Z_S(x) and Z_S's linear coefficient (its formal derivative, vanishing_polynomial.tcc:55-74) for S = span(basis) + shift
int iopx_gf192_vanishing_host(const uint64_t *basis, size_t dim, const uint64_t *shift, const uint64_t *x, uint64_t *value_out, uint64_t *linear_coefficient_out)
{
    if ((dim > 0 && !basis) || !shift || !x) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (dim > 63) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    const SubspacePoly lin(basis, dim);
    if (value_out) { const hgf192 v = lin.eval(hgf192::from_words(x)) + lin.eval(hgf192::from_words(shift)); memcpy(value_out, v.w, 24); }
    if (linear_coefficient_out) memcpy(linear_coefficient_out, lin.coeff[0].w, 24);
    return IOPX_OK;
}

int iopx_gf192_inverse_host(const uint64_t *x, uint64_t *out)
{
    if (!x || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hgf192 v = hgf192::from_words(x);
    if (v.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "inverse of zero");
    const hgf192 r = v.inverse();
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

} // extern "C"
