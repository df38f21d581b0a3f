// LDT reducer: the combined virtual oracle handed to FRI, evaluated over the whole codeword domain on gfx950.
//
// Replaces combined_LDT_virtual_oracle::evaluated_contents (libiop/protocols/ldt/ldt_reducer_aux.tcc:39-131) and the
// subset_element_powers it calls (libiop/algebra/exponentiation.tcc:3-91):
//     result[j] = sum_{maximal k}      c[k] * f_k[j]
//               + sum_{submaximal k}  (c[k] + c[num + i_k] * x_j^(max_degree - degree_k)) * f_k[j]
// with c = {1, random coefficients...} (set_random_coefficients, :26-37) and x_j the j-th domain element.
// One lane owns one position j and walks all oracles, so every codeword is read once and the result written once.
//   * affine subspace: x^(2^i) is GF(2)-linear, so x_j^(2^i) is a subset sum of basis[k]^(2^i) over the bits of j
//     (exponentiation.tcc:3-19); x_j^e is the product over the set bits of e.  The power table is (m + 1) elements per
//     exponent bit; bits of j above the workgroup's 256 positions are wave-uniform.
//   * multiplicative coset: c * shift^e * (g^e)^j from a two-level power table (one product), as the reference's running
//     product (:104-128) yields.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <memory>
#include <vector>
#include "gf192_dev.h"
#include "gf192_host.h"
#include "fp3_dev.h"
#include "fp3_host.h"
#include "runtime.h"

namespace iopx {

struct LdtAddParams {
    const uint64_t *const *oracles; // device array of num_oracles device pointers
    uint64_t *out;
    const uint64_t *tab;            // [bit i][0] = shift^(2^i), [bit i][1 + k] = basis[k]^(2^i); (m + 1) elements per bit
    const uint64_t *coef;           // per oracle: c[k], then the coefficient of its shifted copy (unused when maximal)
    const uint64_t *expo;           // per oracle: max_degree - degree_k (0 = maximal)
    size_t n;
    int m, num_oracles;
};

// x_j^(2^i): shift^(2^i) + sum_{bit k of j} basis[k]^(2^i).  The low 8 bits of j differ across the lanes of a workgroup, the
// rest are uniform.
__device__ __forceinline__ gf192 ldt_subset_sum(const uint64_t *t, int m, uint32_t jlo, uint32_t jhi_uniform)
{
    gf192 v = gf_load(t, 0);
    const int lo_bits = m < 8 ? m : 8;
    for (int k = 0; k < lo_bits; ++k) {
        const uint32_t mask = 0u - ((jlo >> k) & 1u);
        const gf192 b = gf_load(t, 1 + k);
#pragma unroll
        for (int w = 0; w < 6; ++w) v.w[w] = xor_and(v.w[w], mask, b.w[w]);
    }
    for (int k = 8; k < m; ++k) {
        if ((jhi_uniform >> (k - 8)) & 1u) gf_add_to(v, gf_load(t, 1 + k));
    }
    return v;
}

__global__ void __launch_bounds__(256) k_ldt_combine_add(LdtAddParams p)
{
    // a workgroup owns 256 consecutive positions at a time: index bits >= 8 are uniform
    for (size_t base = (size_t)blockIdx.x * 256; base < p.n; base += (size_t)gridDim.x * 256) {
        const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 8));
        const size_t end = base + 256 < p.n ? base + 256 : p.n;
        for (size_t j = base + threadIdx.x; j < end; j += blockDim.x) {
            gf192 acc = gf_zero();
            for (int o = 0; o < p.num_oracles; ++o) {
                const gf192 f = gf_load(p.oracles[o], j);
                gf192 c = gf_load(p.coef, 2 * o);
                uint64_t e = p.expo[o];
                if (e) {
                    gf192 bump = gf_load(p.coef, 2 * o + 1);
                    for (int i = 0; e; ++i, e >>= 1) {
                        if (e & 1) bump = gf_mul(bump, ldt_subset_sum(p.tab + 3 * (size_t)i * (p.m + 1), p.m, (uint32_t)(j & 255), jhi));
                    }
                    gf_add_to(c, bump);
                }
                gf_add_to(acc, gf_mul(c, f));
            }
            gf_store(p.out, j, acc);
        }
    }
}

struct LdtFpParams {
    const uint64_t *const *oracles; // device array of num_oracles device pointers
    const uint64_t *const *hi;      // per oracle: hi table (nullptr = maximal)
    const uint64_t *const *lo;
    uint64_t *out;
    const uint64_t *coef;           // per oracle: c[k]
    size_t n;
    int num_oracles;
};

__global__ void __launch_bounds__(256) k_ldt_combine_fp(LdtFpParams p)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        fp3 acc = fp_zero();
        for (int o = 0; o < p.num_oracles; ++o) {
            const fp3 f = fp_load(p.oracles[o], j);
            fp3 c = fp_load(p.coef, o);
            const uint64_t *hi = p.hi[o];
            if (hi) c = fp_add(c, fp_mul(fp_load(hi, j >> 12), fp_load(p.lo[o], j & 4095)));
            acc = fp_add(acc, fp_mul(c, f));
        }
        fp_store(p.out, j, acc);
    }
}

static int ldt_grid(size_t n)
{
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return (int)(g ? g : 1);
}

// the bookkeeping of combined_LDT_virtual_oracle's constructor + set_random_coefficients (ldt_reducer_aux.tcc:3-37):
// per oracle its own coefficient c[k] (c[0] = 1) and, for the i-th submaximal oracle, c[num + i]
struct LdtPlan {
    size_t max_degree;
    std::vector<size_t> own, shifted;   // indices into the random coefficient array, +1 (0 = the constant one); shifted = 0 when maximal
    std::vector<uint64_t> expo;
};

static int ldt_plan(const size_t *degrees, size_t num, LdtPlan *pl)
{
    if (!degrees || num == 0) return fail(IOPX_ERR_INVALID_ARGUMENT, "Expected same number of evaluations as in registration.");
    pl->max_degree = *std::max_element(degrees, degrees + num);
    size_t sub = 0;
    for (size_t k = 0; k < num; ++k) {
        pl->own.push_back(k);
        if (degrees[k] < pl->max_degree) { pl->shifted.push_back(num + sub); ++sub; }
        else pl->shifted.push_back(0);
        pl->expo.push_back(pl->max_degree - degrees[k]);
    }
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_ldt_combine_gf192_dev(const void *const *d_oracles, size_t num_oracles, const size_t *degrees,
                               const uint64_t *random_coefficients, const uint64_t *basis, size_t m, const uint64_t *shift,
                               uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_oracles || !random_coefficients || !d_out || (m > 0 && !basis) || !shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    LdtPlan pl;
    if ((rc = ldt_plan(degrees, num_oracles, &pl)) != IOPX_OK) return rc;
    // c = {1, random...}: entry t of that list
    auto coef = [&](size_t t) { return t == 0 ? hgf192::one() : hgf192::from_words(random_coefficients + 3 * (t - 1)); };
    std::vector<uint64_t> hcoef;
    uint64_t all = 0;
    for (size_t k = 0; k < num_oracles; ++k) {
        const hgf192 a = coef(pl.own[k]), b = pl.shifted[k] ? coef(pl.shifted[k]) : hgf192::zero();
        hcoef.insert(hcoef.end(), a.w, a.w + 3);
        hcoef.insert(hcoef.end(), b.w, b.w + 3);
        all |= pl.expo[k];
    }
    int nbits = 0;
    while (nbits < 64 && (all >> nbits)) ++nbits;
    // basis[k]^(2^i), shift^(2^i) by repeated squaring (exponentiation.tcc:10-18)
    std::vector<uint64_t> htab((size_t)(nbits ? nbits : 1) * (m + 1) * 3, 0);
    std::vector<hgf192> cur;
    cur.push_back(hgf192::from_words(shift));
    for (size_t k = 0; k < m; ++k) cur.push_back(hgf192::from_words(basis + 3 * k));
    for (int i = 0; i < nbits; ++i) {
        for (size_t k = 0; k <= m; ++k) {
            memcpy(&htab[((size_t)i * (m + 1) + k) * 3], cur[k].w, 24);
            cur[k] = cur[k].squared();
        }
    }
    TmpBuf dptrs, dtab, dcoef, dexpo;
    if ((rc = dptrs.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = dtab.alloc(htab.size() * 8)) != IOPX_OK) return rc;
    if ((rc = dcoef.alloc(hcoef.size() * 8)) != IOPX_OK) return rc;
    if ((rc = dexpo.alloc(num_oracles * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dptrs.p, d_oracles, num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = upload(dtab.p, htab.data(), htab.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dcoef.p, hcoef.data(), hcoef.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dexpo.p, pl.expo.data(), num_oracles * 8)) != IOPX_OK) return rc;
    LdtAddParams p;
    p.oracles = (const uint64_t *const *)dptrs.p;
    p.out = d_out;
    p.tab = dtab.u64(); p.coef = dcoef.u64(); p.expo = dexpo.u64();
    p.n = (size_t)1 << m; p.m = (int)m; p.num_oracles = (int)num_oracles;
    { ProfScope ps_("k_ldt_combine_add"); hipLaunchKernelGGL(k_ldt_combine_add, dim3(ldt_grid(p.n)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_ldt_combine_fp3_dev(const void *const *d_oracles, size_t num_oracles, const size_t *degrees,
                             const uint64_t *random_coefficients, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                             uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_oracles || !random_coefficients || !d_out || !gen || !shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (log_n > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", log_n);
    LdtPlan pl;
    if ((rc = ldt_plan(degrees, num_oracles, &pl)) != IOPX_OK) return rc;
    auto coef = [&](size_t t) { hfp3 r = hfp3::one(); if (t) memcpy(r.w, random_coefficients + 3 * (t - 1), 24); return r; };
    hfp3 g, s;
    memcpy(g.w, gen, 24);
    memcpy(s.w, shift, 24);
    std::vector<uint64_t> hcoef;
    std::vector<std::unique_ptr<TmpBuf>> tabs;
    std::vector<const uint64_t *> hhi(num_oracles, nullptr), hlo(num_oracles, nullptr);
    for (size_t k = 0; k < num_oracles; ++k) {
        const hfp3 a = coef(pl.own[k]);
        hcoef.insert(hcoef.end(), a.w, a.w + 3);
        if (!pl.shifted[k]) continue;
        // cur_bump_factor = c[num + i] * shift^e, multiplied by g^e per position (ldt_reducer_aux.tcc:112-126)
        tabs.emplace_back(new TmpBuf());
        tabs.emplace_back(new TmpBuf());
        TmpBuf &hi = *tabs[tabs.size() - 2], &lo = *tabs[tabs.size() - 1];
        if ((rc = build_two_level(g.pow(pl.expo[k]), coef(pl.shifted[k]) * s.pow(pl.expo[k]), (int)log_n, hi, lo)) != IOPX_OK) return rc;
        hhi[k] = hi.u64();
        hlo[k] = lo.u64();
    }
    TmpBuf dptrs, dhi, dlo, dcoef;
    if ((rc = dptrs.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = dhi.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = dlo.alloc(num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = dcoef.alloc(hcoef.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dptrs.p, d_oracles, num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = upload(dhi.p, hhi.data(), num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = upload(dlo.p, hlo.data(), num_oracles * sizeof(void *))) != IOPX_OK) return rc;
    if ((rc = upload(dcoef.p, hcoef.data(), hcoef.size() * 8)) != IOPX_OK) return rc;
    LdtFpParams p;
    p.oracles = (const uint64_t *const *)dptrs.p;
    p.hi = (const uint64_t *const *)dhi.p;
    p.lo = (const uint64_t *const *)dlo.p;
    p.out = d_out;
    p.coef = dcoef.u64();
    p.n = (size_t)1 << log_n; p.num_oracles = (int)num_oracles;
    { ProfScope ps_("k_ldt_combine_fp"); hipLaunchKernelGGL(k_ldt_combine_fp, dim3(ldt_grid(p.n)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // extern "C"
