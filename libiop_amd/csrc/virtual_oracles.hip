// Virtual-oracle evaluation over the whole codeword domain on gfx950: the R1CS row check and the fz oracle.
//
// Replaces rowcheck_ABC_virtual_oracle::evaluated_contents (libiop/protocols/encoded/common/rowcheck.tcc:16-88):
//     result[x] = Z_H(x)^-1 * (Az(x) * Bz(x) - Cz(x)),     H = the constraint domain.
// Z_H is |H|-to-1 on the codeword domain L, so it takes |L| / |H| values (vanishing_polynomial::unique_evaluations_over_field_subset,
// libiop/algebra/polynomials/vanishing_polynomial.tcc:77-95): the host evaluates and inverts those (32 of them at rate 1/32) and
// the kernel looks the inverse up by coset —
//   affine subspaces: H = span(basis[0..h)) + shift_H is a prefix of L's basis, cosets are contiguous blocks (position >> h);
//   multiplicative cosets: Z_H(x) = x^|H| - shift_H^|H|, the coset of position p is p mod (|L| / |H|) (rowcheck.tcc:50-65).
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "gf192_dev.h"
#include "gf192_host.h"
#include "fp3_dev.h"
#include "fp3_host.h"
#include "runtime.h"

namespace iopx {

__global__ void __launch_bounds__(256) k_rowcheck_add(uint64_t *out, const uint64_t *az, const uint64_t *bz, const uint64_t *cz,
                                                      const uint64_t *zinv, int h, size_t n)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        gf192 t = gf_mul(gf_load(az, j), gf_load(bz, j));
        gf_add_to(t, gf_load(cz, j));                       // characteristic 2: a b - c = a b + c
        // a wavefront covers 64 consecutive positions: with cosets of at least 64 elements 1 / Z_H is the same for all of them
        gf_store(out, j, h >= 6 ? gf_mul_uniform(t, gf_load(zinv, j >> h)) : gf_mul(t, gf_load(zinv, j >> h)));
    }
}

// out[j] = in[j] / Z_H(x_j): one product with the coset's inverse
__global__ void __launch_bounds__(256) k_div_by_vanishing_add(uint64_t *out, const uint64_t *in, const uint64_t *zinv, int h, size_t n)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const gf192 t = gf_load(in, j);
        gf_store(out, j, h >= 6 ? gf_mul_uniform(t, gf_load(zinv, j >> h)) : gf_mul(t, gf_load(zinv, j >> h)));
    }
}

// The field products of this library are data x table (fp3_dev.h): the product of two DATA values comes out as a b 2^181, so
// Cz is brought to the same scale (times the stored 1 = 2^192) and the inverse table carries the missing 2^11 twice.
__global__ void __launch_bounds__(256) k_rowcheck_fp(uint64_t *out, const uint64_t *az, const uint64_t *bz, const uint64_t *cz,
                                                     const uint64_t *zinv_scaled, const uint64_t *one_stored, size_t num_cosets, size_t n)
{
    const fp3 one = fp_load(one_stored, 0);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        fp7w w;                                               // Az Bz - Cz 1 = Az Bz + (p - Cz) 1 with one reduction
        fp7w_zero(w);
        fp_mac(w, fp_load(az, j), fp_load(bz, j));
        fp_mac(w, fp_neg(fp_load(cz, j)), one);
        fp_store(out, j, fp_mul(fp_redc(w), fp_load(zinv_scaled, j & (num_cosets - 1))));
    }
}

// fz_virtual_oracle::evaluated_contents (libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.tcc:181-222):
//     result[x] = fw(x) * Z_I(x) + f_1v(x),      I = the input variable domain, f_1v already extended to the codeword domain.
// Z_I is an affine linearized polynomial, so Z_I(x_j) = Z_I(shift) + sum_{bit k of j} Z_lin(basis[k])
// (vanishing_polynomial::evaluations_over_subspace -> linearized_polynomial.tcc:83-108): an (m + 1)-entry table; index bits
// 0-7 differ across a workgroup's lanes, the rest are uniform.
__global__ void __launch_bounds__(256) k_fz_add(uint64_t *out, const uint64_t *fw, const uint64_t *f1v, const uint64_t *tab, int m, size_t n)
{
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
        const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 8));
        const size_t end = base + 256 < n ? base + 256 : n;
        for (size_t j = base + threadIdx.x; j < end; j += blockDim.x) {
            const gf192 z = subset_sum_ext(tab, m, (uint32_t)(j & 255), jhi);
            gf192 r = gf_mul(gf_load(fw, j), z);
            gf_add_to(r, gf_load(f1v, j));
            gf_store(out, j, r);
        }
    }
}

// multiplicative arm: Z_I(shift g^j) = (shift g^j)^|I| - shift_I^|I|, the power from a two-level table (table form)
__global__ void __launch_bounds__(256) k_fz_fp(uint64_t *out, const uint64_t *fw, const uint64_t *f1v, const uint64_t *hi, const uint64_t *lo,
                                               const uint64_t *vp_shift_t, size_t n)
{
    const fp3 c = fp_load(vp_shift_t, 0);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const fp3 z = fp_sub(fp_mul(fp_load(hi, j >> 12), fp_load(lo, j & 4095)), c);       // table form of Z_I(x_j)
        fp_store(out, j, fp_add(fp_mul(fp_load(fw, j), z), fp_load(f1v, j)));
    }
}

// sumcheck_g_oracle::evaluated_contents (libiop/protocols/encoded/sumcheck/sumcheck.tcc:58-119), affine subspaces:
//     p'(x) = f(x) - eps^-1 mu x^(|H| - 1) - Z_H(x) h(x),      eps = the linear coefficient of Z_H.
// x^(|H| - 1) = x^|H| / x as the reference does (sumcheck_aux.tcc:3-32): x^|H| is a subset sum, the inverses come from
// Montgomery's trick over the 8 positions a lane owns (one field inversion per 8 elements, gf_inv), the constant rides on the
// inverted product for free; a zero element (unshifted domain) inverts to zero (utils.tcc:79-97).
struct SumcheckAddParams {
    const uint64_t *f, *h;
    uint64_t *out;
    const uint64_t *xtab, *htab, *ztab;     // (m + 1)-entry subset-sum tables of x, x^|H|, Z_H(x)
    const uint64_t *c;                      // eps^-1 * mu
    int m;
    size_t n;
};

__device__ __forceinline__ gf192 vo_subset_sum(const uint64_t *t, int m, uint32_t jlo, uint32_t jhi_uniform)
{
    return subset_sum_ext(t, m, jlo, jhi_uniform);
}

#define SUMCHECK_BATCH 8
__global__ void __launch_bounds__(256, 2) k_sumcheck_g_add(SumcheckAddParams p)
{
    // running products of the lane's batch: [r][word][lane] (dynamic r: LDS, not registers)
    __shared__ uint32_t prefix[SUMCHECK_BATCH * 6 * 256];
    const gf192 cst = gf_load(p.c, 0);
    gf192 one = gf_zero();
    one.w[0] = 1;
    // a workgroup owns 256 * SUMCHECK_BATCH consecutive positions; lane t takes t, t + 256, ...: its low 8 index bits are fixed
    for (size_t base = (size_t)blockIdx.x * (256 * SUMCHECK_BATCH); base < p.n; base += (size_t)gridDim.x * (256 * SUMCHECK_BATCH)) {
        for (uint32_t t = threadIdx.x; t < 256; t += blockDim.x) {
            gf192 run = one;
#pragma unroll 1
            for (int r = 0; r < SUMCHECK_BATCH; ++r) {
                const size_t j = base + t + 256 * (size_t)r;
                const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)((base >> 8) + r));
                gf192 x = j < p.n ? vo_subset_sum(p.xtab, p.m, t, jhi) : one;
                if (gf_is_zero(x)) x = one;
                run = gf_mul(run, x);
#pragma unroll
                for (int w = 0; w < 6; ++w) prefix[(r * 6 + w) * 256 + t] = run.w[w];
            }
            gf192 inv = gf_mul(gf_inv(run), cst);
#pragma unroll 1
            for (int r = SUMCHECK_BATCH - 1; r >= 0; --r) {
                const size_t j = base + t + 256 * (size_t)r;
                const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)((base >> 8) + r));
                gf192 x = j < p.n ? vo_subset_sum(p.xtab, p.m, t, jhi) : one;
                const bool zero = gf_is_zero(x);
                if (zero) x = one;
                gf192 before = one;                                 // product of the batch's earlier elements
                if (r > 0) {
#pragma unroll
                    for (int w = 0; w < 6; ++w) before.w[w] = prefix[((r - 1) * 6 + w) * 256 + t];
                }
                gf192 xinv_c = gf_mul(before, inv);                 // c / x
                inv = gf_mul(inv, x);
                if (j >= p.n) continue;
                if (zero) xinv_c = gf_zero();
                gf192 acc = gf_load(p.f, j);
                gf_add_to(acc, gf_mul(vo_subset_sum(p.htab, p.m, t, jhi), xinv_c));
                gf_add_to(acc, gf_mul(vo_subset_sum(p.ztab, p.m, t, jhi), gf_load(p.h, j)));
                gf_store(p.out, j, acc);
            }
        }
    }
}

// mu = 0 (every sum the shipped protocols claim: lincheck attaches its oracle with claimed sum 0, basic_lincheck.tcc:213-214): the
// middle term eps^-1 mu x^(|H| - 1) is identically zero, so p'(x) = f(x) - Z_H(x) h(x) — no inversions at all
__global__ void __launch_bounds__(256) k_sumcheck_g_add_zero_sum(SumcheckAddParams p)
{
    for (size_t base = (size_t)blockIdx.x * 256; base < p.n; base += (size_t)gridDim.x * 256) {
        const uint32_t jhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 8));
        const size_t end = base + 256 < p.n ? base + 256 : p.n;
        for (size_t j = base + threadIdx.x; j < end; j += blockDim.x) {
            gf192 acc = gf_load(p.f, j);
            gf_add_to(acc, gf_mul(vo_subset_sum(p.ztab, p.m, (uint32_t)(j & 255), jhi), gf_load(p.h, j)));
            gf_store(p.out, j, acc);
        }
    }
}

// multiplicative arm (sumcheck.tcc:96-117): p'(x) = (f(x) - |H|^-1 mu - Z_H(x) h(x)) / x
__global__ void __launch_bounds__(256) k_sumcheck_g_fp(uint64_t *out, const uint64_t *f, const uint64_t *h, const uint64_t *zhi, const uint64_t *zlo,
                                                       const uint64_t *ihi, const uint64_t *ilo, const uint64_t *consts, size_t n)
{
    const fp3 vp_shift_t = fp_load(consts, 0), mu_scaled = fp_load(consts, 1);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const fp3 z = fp_sub(fp_mul(fp_load(zhi, j >> 12), fp_load(zlo, j & 4095)), vp_shift_t);
        const fp3 t = fp_sub(fp_sub(fp_load(f, j), mu_scaled), fp_mul(fp_load(h, j), z));
        fp_store(out, j, fp_mul(t, fp_mul(fp_load(ihi, j >> 12), fp_load(ilo, j & 4095))));
    }
}

// multi_lincheck_virtual_oracle::evaluated_contents (libiop/protocols/encoded/lincheck/basic_lincheck_aux.tcc:102-144), given
// p_alpha^1 and p_alpha^2 already extended to the codeword domain (two ordinary transforms, :112-118):
//     result[x] = (sum_m r_m Mz_m(x)) * p_alpha^1(x) - fz(x) * p_alpha^2(x)
#define LINCHECK_MAX_MATRICES 8
struct LincheckParams {
    const uint64_t *fz, *p1, *p2;
    const uint64_t *mz[LINCHECK_MAX_MATRICES];
    const uint64_t *r;              // num_matrices coefficients (fp3: table form), then (fp3 only) the rescaling constant
    uint64_t *out;
    int num_matrices;
    size_t n;
};

__global__ void __launch_bounds__(256) k_lincheck_add(LincheckParams p)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        gf192 comb = gf_zero();
        for (int m = 0; m < p.num_matrices; ++m) gf_add_to(comb, gf_mul_uniform(gf_load(p.mz[m], j), gf_load(p.r, m)));     // r_m: one constant for the launch
        gf192 acc = gf_mul(comb, gf_load(p.p1, j));
        gf_add_to(acc, gf_mul(gf_load(p.fz, j), gf_load(p.p2, j)));
        gf_store(p.out, j, acc);
    }
}

// Both terms are data x data products (scale 2^181, see k_rowcheck_fp): one last product with 2^214 restores libff's form.
__global__ void __launch_bounds__(256) k_lincheck_fp(LincheckParams p)
{
    const fp3 rescale = fp_load(p.r, p.num_matrices);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        fp7w w;                                               // sum_m r_m Mz_m: one reduction (at most 8 matrices)
        fp7w_zero(w);
        for (int m = 0; m < p.num_matrices; ++m) fp_mac(w, fp_load(p.mz[m], j), fp_load(p.r, m));
        const fp3 comb = fp_redc(w);
        fp7w_zero(w);                                         // comb p1 - fz p2 = comb p1 + (p - fz) p2: one reduction
        fp_mac(w, comb, fp_load(p.p1, j));
        fp_mac(w, fp_neg(fp_load(p.fz, j)), fp_load(p.p2, j));
        fp_store(p.out, j, fp_mul(fp_redc(w), rescale));
    }
}

static int vo_grid(size_t n)
{
    size_t g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    return (int)(g ? g : 1);
}

} // namespace iopx

using namespace iopx;

// Montgomery's trick on the host: every element (all non-zero) replaced by its inverse with one field inversion
template<typename H>
static void host_batch_inverse(std::vector<H> &v)
{
    if (v.empty()) return;
    std::vector<H> prefix(v.size());
    H acc = v[0];
    prefix[0] = v[0];
    for (size_t i = 1; i < v.size(); ++i) { acc = acc * v[i]; prefix[i] = acc; }
    H inv = acc.inverse();
    for (size_t i = v.size(); i-- > 1; ) {
        const H vi = v[i];
        v[i] = inv * prefix[i - 1];
        inv = inv * vi;
    }
    v[0] = inv;
}

// 1 / Z_H on each coset of H = span(basis[0..h)) + constraint_shift inside the domain span(basis[0..m)) + shift: a function of the two domains
// only, kept on the device.  Z_H = prod_{v in H} (X - v): the subspace polynomial of span(basis[0..h)) built factor by factor,
// Z <- Z(X) (Z(X) + Z(b)), shifted by its value at shift_H (vanishing_polynomial.tcc:373-395).
static int vanishing_inverse_table(const uint64_t *basis, size_t m, const uint64_t *shift, size_t h, const uint64_t *constraint_shift, TmpBuf &dz)
{
    const size_t cosets = (size_t)1 << (m - h);
    std::vector<uint64_t> key(basis, basis + 3 * m);
    key.insert(key.end(), shift, shift + 3);
    key.insert(key.end(), constraint_shift, constraint_shift + 3);
    key.push_back(m); key.push_back(h); key.push_back(0x726f77);            // "row"
    return cached_domain_table(key, [&](std::vector<uint64_t> &zinv) -> int {
        const std::shared_ptr<CachedSubspacePoly> lin_entry = cached_subspace_poly(basis, h);
        CachedSubspacePoly &lin = *lin_entry;
        auto eval = [&](const hgf192 &x) { return lin.eval(x); };
        const hgf192 z_shift = eval(hgf192::from_words(constraint_shift));
        std::vector<hgf192> z(cosets);
        for (size_t c = 0; c < cosets; ++c) {
            hgf192 x = hgf192::from_words(shift);               // first element of the coset: index c << h (utils.tcc:8-30)
            for (size_t k = h; k < m; ++k) if ((c >> (k - h)) & 1) x += hgf192::from_words(basis + 3 * k);
            z[c] = eval(x) + z_shift;
            if (z[c].is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "the codeword domain intersects the constraint domain");
        }
        host_batch_inverse(z);                                  // one field inversion for all cosets
        zinv.resize(3 * cosets);
        for (size_t c = 0; c < cosets; ++c) memcpy(&zinv[3 * c], z[c].w, 24);
        return IOPX_OK;
    }, dz);
}

extern "C" {

int iopx_div_by_vanishing_gf192_dev(const uint64_t *d_in, const uint64_t *basis, size_t m, const uint64_t *shift, size_t sub_dim, const uint64_t *sub_shift,
                                    uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_in || !d_out || (m > 0 && !basis) || !shift || !sub_shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (sub_dim > m || m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "the vanishing set must be spanned by a prefix of the domain's basis");
    TmpBuf dz;
    if ((rc = vanishing_inverse_table(basis, m, shift, sub_dim, sub_shift, dz)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << m;
    { ProfScope ps_("k_div_by_vanishing_add", 2 * n * 24); hipLaunchKernelGGL(k_div_by_vanishing_add, dim3(vo_grid(n)), dim3(256), 0, stream(), d_out, d_in, (const uint64_t *)dz.u64(), (int)sub_dim, n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_rowcheck_gf192_dev(const uint64_t *d_Az, const uint64_t *d_Bz, const uint64_t *d_Cz, const uint64_t *basis, size_t m,
                            const uint64_t *shift, size_t constraint_dim, const uint64_t *constraint_shift, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_Az || !d_Bz || !d_Cz || !d_out || (m > 0 && !basis) || !shift || !constraint_shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (constraint_dim > m || m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "the constraint domain must be a sub-domain of the codeword domain");
    const size_t h = constraint_dim;
    TmpBuf dz;
    if ((rc = vanishing_inverse_table(basis, m, shift, h, constraint_shift, dz)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << m;
    { ProfScope ps_("k_rowcheck_add", 4 * n * 24); hipLaunchKernelGGL(k_rowcheck_add, dim3(vo_grid(n)), dim3(256), 0, stream(), d_out, d_Az, d_Bz, d_Cz, (const uint64_t *)dz.u64(), (int)h, n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_rowcheck_fp3_dev(const uint64_t *d_Az, const uint64_t *d_Bz, const uint64_t *d_Cz, size_t log_n, const uint64_t *gen,
                          const uint64_t *shift, size_t constraint_log_order, const uint64_t *constraint_shift, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_Az || !d_Bz || !d_Cz || !d_out || !gen || !shift || !constraint_shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (constraint_log_order > log_n || log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "the constraint domain must be a sub-domain of the codeword domain");
    const size_t order_h = (size_t)1 << constraint_log_order, cosets = (size_t)1 << (log_n - constraint_log_order);
    hfp3 g, s, hs;
    memcpy(g.w, gen, 24); memcpy(s.w, shift, 24); memcpy(hs.w, constraint_shift, 24);
    // Z_H(shift g^j) = (shift g^j)^|H| - shift_H^|H| for j < |L| / |H|
    const hfp3 vp_shift = hs.pow(order_h), g_h = g.pow(order_h);
    hfp3 cur = s.pow(order_h);
    std::vector<hfp3> z(cosets);
    for (size_t j = 0; j < cosets; ++j) {
        z[j] = cur - vp_shift;
        if (z[j].is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "the codeword domain intersects the constraint domain");
        cur = cur * g_h;
    }
    host_batch_inverse(z);
    std::vector<uint64_t> zinv(3 * cosets);
    for (size_t j = 0; j < cosets; ++j) {
        const hfp3 zi = z[j].table_form().table_form();
        memcpy(&zinv[3 * j], zi.w, 24);
    }
    const hfp3 one = hfp3::one();
    TmpBuf dz, done;
    if ((rc = dz.alloc(zinv.size() * 8)) != IOPX_OK) return rc;
    if ((rc = done.alloc(24)) != IOPX_OK) return rc;
    if ((rc = upload(dz.p, zinv.data(), zinv.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(done.p, one.w, 24)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << log_n;
    { ProfScope ps_("k_rowcheck_fp", 4 * n * 24); hipLaunchKernelGGL(k_rowcheck_fp, dim3(vo_grid(n)), dim3(256), 0, stream(), d_out, d_Az, d_Bz, d_Cz, (const uint64_t *)dz.u64(),
                                                        (const uint64_t *)done.u64(), cosets, n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_fz_gf192_dev(const uint64_t *d_fw, const uint64_t *d_f1v, const uint64_t *basis, size_t m, const uint64_t *shift,
                      const uint64_t *input_basis, size_t input_dim, const uint64_t *input_shift, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_fw || !d_f1v || !d_out || (m > 0 && !basis) || !shift || (input_dim > 0 && !input_basis) || !input_shift)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (input_dim > m || m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "Codeword domain must be bigger than the input variable domain.");
    // the subset-sum table of Z_I over the codeword domain: a function of the two domains only, kept on the device
    std::vector<uint64_t> key(basis, basis + 3 * m);
    key.insert(key.end(), shift, shift + 3);
    key.insert(key.end(), input_basis, input_basis + 3 * input_dim);
    key.insert(key.end(), input_shift, input_shift + 3);
    key.push_back(m); key.push_back(input_dim); key.push_back(0x667a);      // "fz"
    TmpBuf dt;
    rc = cached_domain_table(key, [&](std::vector<uint64_t> &tab) -> int {
        const std::shared_ptr<CachedSubspacePoly> lin_entry = cached_subspace_poly(input_basis, input_dim);       // Z_I's linear part
        CachedSubspacePoly &lin = *lin_entry;
        auto eval = [&](const hgf192 &x) { return lin.eval(x); };
        std::vector<hgf192> entries;
        entries.push_back(eval(hgf192::from_words(shift)) + eval(hgf192::from_words(input_shift)));    // Z_I(shift) = lin(shift) + lin(shift_I)
        for (size_t k = 0; k < m; ++k) entries.push_back(eval(hgf192::from_words(basis + 3 * k)));
        append_subset_table_with_ext(tab, entries);
        return IOPX_OK;
    }, dt);
    if (rc != IOPX_OK) return rc;
    const size_t n = (size_t)1 << m;
    { ProfScope ps_("k_fz_add", 3 * n * 24); hipLaunchKernelGGL(k_fz_add, dim3(vo_grid(n)), dim3(256), 0, stream(), d_out, d_fw, d_f1v, (const uint64_t *)dt.u64(), (int)m, n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_fz_fp3_dev(const uint64_t *d_fw, const uint64_t *d_f1v, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                    size_t input_log_order, const uint64_t *input_shift, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_fw || !d_f1v || !d_out || !gen || !shift || !input_shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (input_log_order > log_n || log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "Codeword domain must be bigger than the input variable domain.");
    hfp3 g, s, is;
    memcpy(g.w, gen, 24); memcpy(s.w, shift, 24); memcpy(is.w, input_shift, 24);
    const uint64_t order_i = (uint64_t)1 << input_log_order;
    TmpBuf hi, lo, dc;
    if ((rc = build_two_level(g.pow(order_i), s.pow(order_i), (int)log_n, hi, lo)) != IOPX_OK) return rc;
    const hfp3 c = is.pow(order_i).table_form();
    if ((rc = dc.alloc(24)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, c.w, 24)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << log_n;
    { ProfScope ps_("k_fz_fp", 3 * n * 24); hipLaunchKernelGGL(k_fz_fp, dim3(vo_grid(n)), dim3(256), 0, stream(), d_out, d_fw, d_f1v, (const uint64_t *)hi.u64(),
                                                  (const uint64_t *)lo.u64(), (const uint64_t *)dc.u64(), n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_sumcheck_g_gf192_dev(const uint64_t *d_f, const uint64_t *d_h, const uint64_t *basis, size_t m, const uint64_t *shift,
                              const uint64_t *summation_basis, size_t summation_dim, const uint64_t *summation_shift,
                              const uint64_t *claimed_sum, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_f || !d_h || !d_out || (m > 0 && !basis) || !shift || (summation_dim > 0 && !summation_basis) || !summation_shift || !claimed_sum)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (m > 40 || summation_dim > 63) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    const std::shared_ptr<CachedSubspacePoly> lin_entry = cached_subspace_poly(summation_basis, summation_dim);   // Z_H's linear part
    CachedSubspacePoly &lin_cached = *lin_entry;
    const SubspacePoly &lin = lin_cached.poly;
    auto eval = [&](const hgf192 &x) { return lin_cached.eval(x); };
    if (lin.coeff[0].is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "the summation domain's basis is linearly dependent");
    const hgf192 c = lin.coeff[0].inverse() * hgf192::from_words(claimed_sum);   // eps^-1 mu (sumcheck.tcc:52-54)
    // the subset-sum tables of x, x^|H| and Z_H(x) over the codeword domain: functions of the two domains only, kept on the device (one block)
    std::vector<uint64_t> key(basis, basis + 3 * m);
    key.insert(key.end(), shift, shift + 3);
    key.insert(key.end(), summation_basis, summation_basis + 3 * summation_dim);
    key.insert(key.end(), summation_shift, summation_shift + 3);
    key.push_back(m); key.push_back(summation_dim); key.push_back(0x73756d);     // "sum"
    const size_t table_words = 3 * (m + 1) + SUBSET_TABLE_WORDS_EXTRA;
    TmpBuf dtabs, dc;
    rc = cached_domain_table(key, [&](std::vector<uint64_t> &words) -> int {
        std::vector<hgf192> xe, he, ze;
        for (size_t k = 0; k <= m; ++k) {
            const hgf192 v = hgf192::from_words(k == 0 ? shift : basis + 3 * (k - 1));
            hgf192 vh = v;
            for (size_t i = 0; i < summation_dim; ++i) vh = vh.squared();             // v^|H|
            hgf192 vz = eval(v);
            if (k == 0) vz += eval(hgf192::from_words(summation_shift));             // Z_H(shift) = lin(shift) + lin(shift_H)
            xe.push_back(v); he.push_back(vh); ze.push_back(vz);
        }
        append_subset_table_with_ext(words, xe);
        append_subset_table_with_ext(words, he);
        append_subset_table_with_ext(words, ze);
        return words.size() == 3 * table_words ? IOPX_OK : fail(IOPX_ERR_LOGIC, "subset table size");
    }, dtabs);
    if (rc != IOPX_OK) return rc;
    if ((rc = dc.alloc(24)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, c.w, 24)) != IOPX_OK) return rc;
    struct { const uint64_t *p; const uint64_t *u64() const { return p; } } dx{ dtabs.u64() }, dh{ dtabs.u64() + table_words }, dz{ dtabs.u64() + 2 * table_words };
    SumcheckAddParams p;
    p.f = d_f; p.h = d_h; p.out = d_out;
    p.xtab = dx.u64(); p.htab = dh.u64(); p.ztab = dz.u64(); p.c = dc.u64();
    p.m = (int)m; p.n = (size_t)1 << m;
    if (c.is_zero()) {
        { ProfScope ps_("k_sumcheck_g_add_zero_sum", 3 * p.n * 24); hipLaunchKernelGGL(k_sumcheck_g_add_zero_sum, dim3(vo_grid(p.n)), dim3(256), 0, stream(), p); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    }
    size_t g = (p.n + 256 * SUMCHECK_BATCH - 1) / (256 * SUMCHECK_BATCH);
    if (g > 16384) g = 16384;
    { ProfScope ps_("k_sumcheck_g_add"); hipLaunchKernelGGL(k_sumcheck_g_add, dim3((unsigned)(g ? g : 1)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_sumcheck_g_fp3_dev(const uint64_t *d_f, const uint64_t *d_h, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                            size_t summation_log_order, const uint64_t *summation_shift, const uint64_t *claimed_sum, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_f || !d_h || !d_out || !gen || !shift || !summation_shift || !claimed_sum) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (summation_log_order > 62 || log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    hfp3 g, s, hs, mu;
    memcpy(g.w, gen, 24); memcpy(s.w, shift, 24); memcpy(hs.w, summation_shift, 24); memcpy(mu.w, claimed_sum, 24);
    if (s.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "zero coset shift");
    const uint64_t order_h = (uint64_t)1 << summation_log_order;
    TmpBuf zhi, zlo, ihi, ilo, dc;
    if ((rc = build_two_level(g.pow(order_h), s.pow(order_h), (int)log_n, zhi, zlo)) != IOPX_OK) return rc;
    if ((rc = build_two_level(g.inverse(), s.inverse(), (int)log_n, ihi, ilo)) != IOPX_OK) return rc;
    const hfp3 vp_shift_t = hs.pow(order_h).table_form();
    const hfp3 mu_scaled = hfp3::from_uint(order_h).inverse() * mu;              // |H|^-1 mu, a data value (sumcheck.tcc:46-49)
    uint64_t consts[6];
    memcpy(consts, vp_shift_t.w, 24); memcpy(consts + 3, mu_scaled.w, 24);
    if ((rc = dc.alloc(48)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, consts, 48)) != IOPX_OK) return rc;
    const size_t n = (size_t)1 << log_n;
    { ProfScope ps_("k_sumcheck_g_fp", 3 * n * 24); hipLaunchKernelGGL(k_sumcheck_g_fp, dim3(vo_grid(n)), dim3(256), 0, stream(), d_out, d_f, d_h, (const uint64_t *)zhi.u64(),
                                                          (const uint64_t *)zlo.u64(), (const uint64_t *)ihi.u64(), (const uint64_t *)ilo.u64(), (const uint64_t *)dc.u64(), n); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

static int lincheck_common(const uint64_t *d_fz, const void *const *d_Mz, size_t num_matrices, const uint64_t *r_Mz, const uint64_t *d_p1,
                           const uint64_t *d_p2, size_t n, uint64_t *d_out, bool prime_field)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_fz || !d_Mz || !r_Mz || !d_p1 || !d_p2 || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (num_matrices == 0 || num_matrices > LINCHECK_MAX_MATRICES)
        return fail(IOPX_ERR_INVALID_ARGUMENT, "multi_lincheck uses more constituent oracles than what was provided.");
    std::vector<uint64_t> hr(r_Mz, r_Mz + 3 * num_matrices);
    if (prime_field) {
        for (size_t m = 0; m < num_matrices; ++m) { const hfp3 t = hfp3::from_words(r_Mz + 3 * m).table_form(); memcpy(&hr[3 * m], t.w, 24); }
        const hfp3 k = hfp3::one().table_form().table_form();
        hr.insert(hr.end(), k.w, k.w + 3);
    }
    TmpBuf dr;
    if ((rc = dr.alloc(hr.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dr.p, hr.data(), hr.size() * 8)) != IOPX_OK) return rc;
    LincheckParams p;
    memset(&p, 0, sizeof(p));
    p.fz = d_fz; p.p1 = d_p1; p.p2 = d_p2; p.out = d_out; p.r = dr.u64();
    for (size_t m = 0; m < num_matrices; ++m) p.mz[m] = (const uint64_t *)d_Mz[m];
    p.num_matrices = (int)num_matrices; p.n = n;
    if (prime_field) { ProfScope ps_("k_lincheck_fp", (num_matrices + 4) * n * 24); hipLaunchKernelGGL(k_lincheck_fp, dim3(vo_grid(n)), dim3(256), 0, stream(), p); }
    else { ProfScope ps_("k_lincheck_add", (num_matrices + 4) * n * 24); hipLaunchKernelGGL(k_lincheck_add, dim3(vo_grid(n)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_lincheck_gf192_dev(const uint64_t *d_fz, const void *const *d_Mz, size_t num_matrices, const uint64_t *r_Mz,
                            const uint64_t *d_p_alpha_prime, const uint64_t *d_p_alpha_ABC, size_t n, uint64_t *d_out)
{
    return lincheck_common(d_fz, d_Mz, num_matrices, r_Mz, d_p_alpha_prime, d_p_alpha_ABC, n, d_out, false);
}

int iopx_lincheck_fp3_dev(const uint64_t *d_fz, const void *const *d_Mz, size_t num_matrices, const uint64_t *r_Mz,
                          const uint64_t *d_p_alpha_prime, const uint64_t *d_p_alpha_ABC, size_t n, uint64_t *d_out)
{
    return lincheck_common(d_fz, d_Mz, num_matrices, r_Mz, d_p_alpha_prime, d_p_alpha_ABC, n, d_out, true);
}

} // extern "C"
