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
#include <map>
#include <memory>
#include <mutex>
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
    return subset_sum_ext(t, m, jlo, jhi_uniform);
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
                        if (e & 1) bump = gf_mul(bump, ldt_subset_sum(p.tab + 3 * (size_t)i * (p.m + 1 + SUBSET_EXT_ENTRIES), p.m, (uint32_t)(j & 255), jhi));
                    }
                    gf_add_to(c, bump);
                }
                gf_add_to(acc, gf_mul(c, f));
            }
            gf_store(p.out, j, acc);
        }
    }
}

// Shared powers: many oracles have the same degree, and the exponents 2^k - 1 - small of an Aurora-style degree spread share
// most of their bits.  The host turns the distinct exponents into "slots" (a bit set, optionally on top of a parent slot's
// value); a lane computes every slot once into LDS and each oracle picks its slot.
#define LDT_MAX_SLOTS 16
struct LdtSlotParams {
    LdtAddParams a;
    const uint64_t *slot_bits;      // per slot: exponent bits multiplied onto the parent's value
    const int *slot_parent;         // per slot: parent slot or -1
    const int *oracle_slot;         // per oracle: slot or -1 (maximal)
    int num_slots;
    // Degree gap 1 over a domain of one-word points (the standard-basis subspaces libiop builds: a point is a polynomial of degree < 32).  The
    // oracles of that gap are summed first: sum_k (c_k + c'_k x) f_k = sum_k c_k f_k + x sum_k c'_k f_k — two comb products per oracle (the
    // coefficients are wave-uniform) and ONE product by the one-word x for the group (six word products) instead of a comb and a general product
    // per oracle.  In an Aurora proof three of the four sub-maximal oracles have gap 1 (h, g, the row check): 7 + 4 -> 10 + 1 + 1 products.
    int small_slot;                 // the slot of exponent 1 handled that way, or -1
    uint32_t small_shift, small_basis[32];      // x_j = small_shift + sum_{bit k of j} small_basis[k]
};

__global__ void __launch_bounds__(256) k_ldt_combine_add_slots(LdtSlotParams q)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint32_t *iopx_ldt_smem = (uint32_t *)iopx_smem;                            // [slot][word][lane]
    const LdtAddParams &p = q.a;
    for (size_t base = (size_t)blockIdx.x * 256; base < p.n; base += (size_t)gridDim.x * 256) {
        const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 8));
        const size_t end = base + 256 < p.n ? base + 256 : p.n;
        for (size_t j = base + threadIdx.x; j < end; j += blockDim.x) {
            const uint32_t lane = (uint32_t)(j & 255);
            for (int s = 0; s < q.num_slots; ++s) {
                const int parent = q.slot_parent[s];
                uint64_t e = q.slot_bits[s];
                gf192 v;
                bool have = parent >= 0;
                if (have) {
#pragma unroll
                    for (int w = 0; w < 6; ++w) v.w[w] = iopx_ldt_smem[(parent * 6 + w) * 256 + lane];
                }
                for (int i = 0; e; ++i, e >>= 1) {
                    if (!(e & 1)) continue;
                    const gf192 t = ldt_subset_sum(p.tab + 3 * (size_t)i * (p.m + 1 + SUBSET_EXT_ENTRIES), p.m, lane, jhi);
                    v = have ? gf_mul(v, t) : t;
                    have = true;
                }
#pragma unroll
                for (int w = 0; w < 6; ++w) iopx_ldt_smem[(s * 6 + w) * 256 + lane] = v.w[w];
            }
            gf192 acc = gf_zero(), gap1 = gf_zero();
            for (int o = 0; o < p.num_oracles; ++o) {
                const gf192 f = gf_load(p.oracles[o], j);
                gf192 c = gf_load(p.coef, 2 * o);
                const int s = q.oracle_slot[o];
                if (s < 0) {                                 // an oracle of maximal degree: its multiplier is one constant — comb product
                    gf_add_to(acc, gf_mul_uniform(f, c));
                    continue;
                }
                if (s == q.small_slot) {                     // gap 1, one-word points: both coefficients by the comb, x once for the group below
                    gf_add_to(acc, gf_mul_uniform(f, c));
                    gf_add_to(gap1, gf_mul_uniform(f, gf_load(p.coef, 2 * o + 1)));
                    continue;
                }
                gf192 v;
#pragma unroll
                for (int w = 0; w < 6; ++w) v.w[w] = iopx_ldt_smem[(s * 6 + w) * 256 + lane];
                gf_add_to(c, gf_mul_uniform(v, gf_load(p.coef, 2 * o + 1)));         // the coefficient is the same for every lane: comb product
                gf_add_to(acc, gf_mul(c, f));
            }
            if (q.small_slot >= 0) {
                uint32_t x = q.small_shift;
                for (int k = 0; k < p.m; ++k) x ^= (0u - (uint32_t)((j >> k) & 1)) & q.small_basis[k];
                gf_add_to(acc, gf_mul_small_over_xk(gap1, x, 0));
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
        for (int g = 0; g < p.num_oracles; g += 8) {          // sum_o c_o(x) f_o(x): up to 8 products per reduction (fp3_dev.h)
            fp7w w;
            fp7w_zero(w);
            const int e = g + 8 < p.num_oracles ? g + 8 : p.num_oracles;
            for (int o = g; o < e; ++o) {
                fp3 c = fp_load(p.coef, o);
                const uint64_t *hi = p.hi[o];
                if (hi) c = fp_add(c, fp_mul(fp_load(hi, j >> 12), fp_load(p.lo[o], j & 4095)));
                fp_mac(w, c, fp_load(p.oracles[o], j));
            }
            acc = fp_add(acc, fp_redc(w));
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
    // basis[k]^(2^i), shift^(2^i) by repeated squaring (exponentiation.tcc:10-18).  The tables depend on the domain and on which exponent bits occur,
    // not on the proof: kept on the device per (basis, shift, bit mask) — building them (m squarings and 1024 pre-summed entries per bit) and
    // uploading ~25 KB per bit sat on the critical path between two rounds of every proof.
    TmpBuf dptrs, dtab, dcoef, dexpo;
    {
        std::vector<uint64_t> key(basis, basis + 3 * m);
        key.insert(key.end(), shift, shift + 3);
        key.push_back(m); key.push_back(all); key.push_back(0x6c6474);      // "ldt"
        rc = cached_domain_table(key, [&](std::vector<uint64_t> &htab) -> int {     // table i: x^(2^i) over the domain, in the subset-sum layout
            std::vector<hgf192> cur;
            cur.push_back(hgf192::from_words(shift));
            for (size_t k = 0; k < m; ++k) cur.push_back(hgf192::from_words(basis + 3 * k));
            for (int i = 0; i < (nbits ? nbits : 1); ++i) {
                if ((all >> i) & 1) append_subset_table_with_ext(htab, cur);
                else htab.resize(htab.size() + 3 * (m + 1) + SUBSET_TABLE_WORDS_EXTRA, 0);  // never read: no exponent has this bit
                for (size_t k = 0; k <= m; ++k) cur[k] = cur[k].squared();
            }
            return IOPX_OK;
        }, dtab);
        if (rc != IOPX_OK) return rc;
    }
    LdtAddParams p;
    p.out = d_out;
    p.tab = dtab.u64();
    p.n = (size_t)1 << m; p.m = (int)m; p.num_oracles = (int)num_oracles;
    // slots: distinct exponents; those with >= 3 bits that all contain a common bit set hang off one slot holding that set
    std::vector<uint64_t> distinct;
    for (uint64_t e : pl.expo) if (e && std::find(distinct.begin(), distinct.end(), e) == distinct.end()) distinct.push_back(e);
    // x^e costs popcount(e) - 1 products per element.  The heavy exponents (at least half as many bits as the heaviest: gaps of the
    // form 2^k - small, e.g. 2^21 - 3 and 2^21 - 2 in a Fractal proof) share most of their bits: that common part is one parent
    // slot computed once; light exponents stand alone.
    int max_pop = 0;
    for (uint64_t e : distinct) max_pop = std::max(max_pop, __builtin_popcountll(e));
    const int heavy_pop = std::max(3, (max_pop + 1) / 2);
    uint64_t common = ~(uint64_t)0;
    size_t multi = 0;
    for (uint64_t e : distinct) if (__builtin_popcountll(e) >= heavy_pop) { common &= e; ++multi; }
    if (multi < 2 || __builtin_popcountll(common) < 2) common = 0;
    std::vector<uint64_t> slot_bits;
    std::vector<int> slot_parent, oracle_slot(num_oracles, -1);
    if (common) { slot_bits.push_back(common); slot_parent.push_back(-1); }
    for (uint64_t e : distinct) {
        int slot;
        if (common && e == common) slot = 0;
        else {
            const bool on_common = common && __builtin_popcountll(e) >= heavy_pop;
            slot_bits.push_back(on_common ? e & ~common : e);
            slot_parent.push_back(on_common ? 0 : -1);
            slot = (int)slot_bits.size() - 1;
        }
        for (size_t k = 0; k < num_oracles; ++k) if (pl.expo[k] == e) oracle_slot[k] = slot;
    }
    // the per-call tables travel in ONE block (one constant-carrying launch): oracle pointers, coefficient pairs, exponents, then the slot tables
    const bool slots = slot_bits.size() <= LDT_MAX_SLOTS;
    const size_t ns = slot_bits.size() ? slot_bits.size() : 1;
    if (slots) { slot_bits.resize(ns, 0); slot_parent.resize(ns, -1); }
    std::vector<uint64_t> meta;
    const size_t off_ptrs = 0, off_coef = num_oracles, off_expo = off_coef + hcoef.size(), off_bits = off_expo + num_oracles,
                 off_parent = off_bits + (slots ? ns : 0), off_slot = off_parent + (slots ? (ns + 1) / 2 : 0), total = off_slot + (slots ? (num_oracles + 1) / 2 : 0);
    meta.resize(total, 0);
    for (size_t k = 0; k < num_oracles; ++k) meta[off_ptrs + k] = (uint64_t)(uintptr_t)d_oracles[k];
    std::memcpy(&meta[off_coef], hcoef.data(), hcoef.size() * 8);
    std::memcpy(&meta[off_expo], pl.expo.data(), num_oracles * 8);
    if (slots) {
        std::memcpy(&meta[off_bits], slot_bits.data(), ns * 8);
        std::memcpy(&meta[off_parent], slot_parent.data(), ns * 4);
        std::memcpy(&meta[off_slot], oracle_slot.data(), num_oracles * 4);
    }
    if ((rc = dptrs.alloc(total * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dptrs.p, meta.data(), total * 8)) != IOPX_OK) return rc;
    const uint64_t *dmeta = dptrs.u64();
    p.oracles = (const uint64_t *const *)(dmeta + off_ptrs);
    p.coef = dmeta + off_coef; p.expo = dmeta + off_expo;
    if (!slots) {
        ProfScope ps_("k_ldt_combine_add");
        hipLaunchKernelGGL(k_ldt_combine_add, dim3(ldt_grid(p.n)), dim3(256), 0, stream(), p);
    } else {
        LdtSlotParams q;
        memset(&q, 0, sizeof(q));
        q.small_slot = -1;
        {
            // the slot of exponent 1, when every point of the domain is a one-word polynomial (basis vectors and shift below 2^32) and the
            // slot is a plain one (no parent, nobody's parent)
            auto one_word = [](const uint64_t *w) { return w[1] == 0 && w[2] == 0 && (w[0] >> 32) == 0; };
            bool small = m <= 32 && one_word(shift);
            for (size_t k = 0; small && k < m; ++k) small = one_word(basis + 3 * k);
            if (small) {
                for (size_t sidx = 0; sidx < slot_bits.size(); ++sidx)
                    if (slot_bits[sidx] == 1 && slot_parent[sidx] < 0 && std::find(slot_parent.begin(), slot_parent.end(), (int)sidx) == slot_parent.end()) q.small_slot = (int)sidx;
                q.small_shift = (uint32_t)shift[0];
                for (size_t k = 0; k < m; ++k) q.small_basis[k] = (uint32_t)basis[3 * k];
            }
        }
        q.a = p;
        q.slot_bits = dmeta + off_bits; q.slot_parent = (const int *)(dmeta + off_parent); q.oracle_slot = (const int *)(dmeta + off_slot);
        q.num_slots = (int)(distinct.empty() ? 0 : ns);
        const size_t lds_bytes = ns * 6 * 256 * 4;           // 11..16 distinct degree gaps exceed the 64 KiB default
        if (lds_bytes > 64 * 1024)
            IOPX_HIP(hipFuncSetAttribute((const void *)k_ldt_combine_add_slots, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        ProfScope ps_("k_ldt_combine_add_slots", (p.num_oracles + 1) * p.n * 24);
        hipLaunchKernelGGL(k_ldt_combine_add_slots, dim3(ldt_grid(p.n)), dim3(256), lds_bytes, stream(), q);
    }
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
        const hfp3 a = coef(pl.own[k]).table_form();       // multipliers are uploaded in the device's table form (fp3_dev.h)
        hcoef.insert(hcoef.end(), a.w, a.w + 3);
        if (!pl.shifted[k]) continue;
        // cur_bump_factor = c[num + i] * shift^e, multiplied by g^e per position (ldt_reducer_aux.tcc:112-126)
        tabs.emplace_back(new TmpBuf());
        tabs.emplace_back(new TmpBuf());
        TmpBuf &hi = *tabs[tabs.size() - 2], &lo = *tabs[tabs.size() - 1];
        // the random coefficient rides on the hi half (rebuilt per call); the lo half depends on the degree gap and the domain only
        if ((rc = build_two_level(g.pow(pl.expo[k]), coef(pl.shifted[k]) * s.pow(pl.expo[k]), (int)log_n, hi, lo, false)) != IOPX_OK) return rc;
        hhi[k] = hi.u64();
        hlo[k] = lo.u64();
    }
    // one block for the four per-call tables (one constant-carrying launch): oracle pointers, the two table-pointer lists, the coefficients
    std::vector<uint64_t> meta(3 * num_oracles + hcoef.size());
    for (size_t k = 0; k < num_oracles; ++k) {
        meta[k] = (uint64_t)(uintptr_t)d_oracles[k];
        meta[num_oracles + k] = (uint64_t)(uintptr_t)hhi[k];
        meta[2 * num_oracles + k] = (uint64_t)(uintptr_t)hlo[k];
    }
    std::memcpy(&meta[3 * num_oracles], hcoef.data(), hcoef.size() * 8);
    TmpBuf dmeta;
    if ((rc = dmeta.alloc(meta.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dmeta.p, meta.data(), meta.size() * 8)) != IOPX_OK) return rc;
    LdtFpParams p;
    p.oracles = (const uint64_t *const *)dmeta.u64();
    p.hi = (const uint64_t *const *)(dmeta.u64() + num_oracles);
    p.lo = (const uint64_t *const *)(dmeta.u64() + 2 * num_oracles);
    p.out = d_out;
    p.coef = dmeta.u64() + 3 * num_oracles;
    p.n = (size_t)1 << log_n; p.num_oracles = (int)num_oracles;
    { ProfScope ps_("k_ldt_combine_fp", (num_oracles + 1) * p.n * 24); hipLaunchKernelGGL(k_ldt_combine_fp, dim3(ldt_grid(p.n)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // extern "C"
