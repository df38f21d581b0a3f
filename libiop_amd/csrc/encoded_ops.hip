// The vector-sized steps of the encoded Aurora prover that are not transforms or virtual oracles, on gfx950:
//
//   iopx_spmv_*                  r1cs_constraint_system::create_Az_Bz_Cz_from_variable_assignment (libiop/relations/r1cs.tcc:236-268)
//                                and the p_alpha_ABC accumulation of multi_lincheck_virtual_oracle::set_challenge
//                                (libiop/protocols/encoded/lincheck/basic_lincheck_aux.tcc:64-88), both as CSR row gathers
//   iopx_poly_div_vanishing_*    polynomial_over_vanishing_polynomial(...).first (libiop/algebra/polynomials/
//                                vanishing_polynomial.tcc:314-371, linearized_polynomial.tcc:238-289): f_w = f_w' / Z_I
//                                (r1cs_rs_iop.tcc:563-565) and the sumcheck's h = f / Z_H (sumcheck.tcc:359-365)
//   iopx_lincomb_*               random_linear_combination_oracle::evaluated_contents (encoded/common/random_linear_combination.tcc:27-57)
//   iopx_*_{add,sub,mul,inv}_dev elementwise field helpers (the synthetic-instance generator r1cs_examples.tcc:40-64, f_w' = z - f_1v)
//
// Division without the reference's serial sweep.  With n coefficients, N = deg Z and M = n - N quotient coefficients, reversing
// the polynomials turns P = Q Z + R into rev(Q) = rev(P) / rev(Z) mod Y^M, rev(Z) = 1 + u(Y) with u sparse:
//   subspaces: u = sum_i c_i Y^(N - 2^i) (+ c_const Y^N),   cosets: u = -c Y^N.
// (1 + u)^-1 = prod_k (1 + u^(2^k)) in characteristic 2 (Frobenius keeps u^(2^k) as sparse as u); (1 - x)^-1 = prod_k (1 + x^(2^k))
// in any field.  Each factor is one data-parallel pass Q[j] += sum_t const_t * Q[j + offset_t]; the offsets double per pass, so
// log2(M / (N/2)) passes suffice (one pass for the sumcheck's h, where deg f < 2 |H|).  Quotients are unique: same bytes.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "gf192_dev.h"
#include "gf192_host.h"
#include "fp3_dev.h"
#include "fp3_host.h"
#include "runtime.h"

namespace iopx {

static int eo_grid(size_t n)
{
    size_t g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    return (int)(g ? g : 1);
}

// ---- elementwise ----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gf192_add(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t n)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        gf_store(out, j, gf_add(gf_load(a, j), gf_load(b, j)));
}

__global__ void __launch_bounds__(256) k_gf192_inv(uint64_t *out, const uint64_t *a, size_t n)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const gf192 x = gf_load(a, j);
        gf_store(out, j, gf_is_zero(x) ? x : gf_inv(x));
    }
}

// F_p: the device product is a b 2^-203 (fp3_dev.h).  Data live as x 2^192 (libff), so data x data = x y 2^181 and one more
// product with the raw constant 2^214 restores libff's form; consts[0] = 2^214, consts[1] = 2^192 (raw words).
__global__ void __launch_bounds__(256) k_fp3_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, const uint64_t *consts, size_t n)
{
    const fp3 k214 = fp_load(consts, 0);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        fp_store(out, j, fp_mul(fp_mul(fp_load(a, j), fp_load(b, j)), k214));
}

__global__ void __launch_bounds__(256) k_fp3_sub(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t n)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        fp_store(out, j, fp_sub(fp_load(a, j), fp_load(b, j)));
}

// x^(p-2) in the 2^203 form (closed under fp_mul), exponent bits in consts[2] (three raw words), back to libff's form at the end
__global__ void __launch_bounds__(256) k_fp3_inv(uint64_t *out, const uint64_t *a, const uint64_t *consts, size_t n)
{
    const fp3 k214 = fp_load(consts, 0), k192 = fp_load(consts, 1);
    const uint64_t e0 = consts[6], e1 = consts[7], e2 = consts[8];
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const fp3 x = fp_mul(fp_load(a, j), k214);          // x 2^203
        fp3 r = x;                                          // the exponent's top bit (bit 180) is set
        for (int bit = 179; bit >= 0; --bit) {
            r = fp_mul(r, r);
            const uint64_t w = bit >= 128 ? e2 : (bit >= 64 ? e1 : e0);
            if ((w >> (bit & 63)) & 1) r = fp_mul(r, x);
        }
        fp_store(out, j, fp_mul(r, k192));
    }
}

// ---- random linear combination ---------------------------------------------------------------------------------------------
#define LINCOMB_MAX 16
struct LincombParams {
    const uint64_t *o[LINCOMB_MAX];
    const uint64_t *c;          // num coefficients (fp3: 2^203 form), then the constant term (libff's form) when has_constant
    uint64_t *out;
    int num, has_constant;
    size_t n;
};

__global__ void __launch_bounds__(256) k_lincomb_gf192(LincombParams p)
{
    const gf192 c0 = p.has_constant ? gf_load(p.c, p.num) : gf_zero();
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        gf192 acc = c0;
        for (int i = 0; i < p.num; ++i) gf_add_to(acc, gf_mul_uniform(gf_load(p.o[i], j), gf_load(p.c, i)));
        gf_store(p.out, j, acc);
    }
}

__global__ void __launch_bounds__(256) k_lincomb_fp3(LincombParams p)
{
    const fp3 c0 = p.has_constant ? fp_load(p.c, p.num) : fp_zero();
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += (size_t)gridDim.x * blockDim.x) {
        fp3 acc = c0;
        for (int g = 0; g < p.num; g += 8) {                 // up to 8 products per reduction (fp3_dev.h)
            fp7w w;
            fp7w_zero(w);
            const int e = g + 8 < p.num ? g + 8 : p.num;
            for (int i = g; i < e; ++i) fp_mac(w, fp_load(p.o[i], j), fp_load(p.c, i));
            acc = fp_add(acc, fp_redc(w));
        }
        fp_store(p.out, j, acc);
    }
}

// ---- CSR sparse matrix x vector: one lane per row ----------------------------------------------------------------------------
struct SpmvParams {
    const uint64_t *row_ptr;    // rows + 1 offsets
    const uint32_t *col;
    const uint64_t *coeff, *vec;
    const uint64_t *scale;      // device: one element (fp3: r 2^214 raw, or 2^214 when there is no scale); gf192: nullable
    uint64_t *out;
    size_t rows;
    int accumulate;
};

__global__ void __launch_bounds__(256) k_spmv_gf192(SpmvParams p)
{
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.rows; r += (size_t)gridDim.x * blockDim.x) {
        gf192 acc = gf_zero();
        const uint64_t e = p.row_ptr[r + 1];
        for (uint64_t t = p.row_ptr[r]; t < e; ++t) gf_add_to(acc, gf_mul(gf_load(p.vec, p.col[t]), gf_load(p.coeff, t)));
        if (p.scale) acc = gf_mul(acc, gf_load(p.scale, 0));
        if (p.accumulate) gf_add_to(acc, gf_load(p.out, r));
        gf_store(p.out, r, acc);
    }
}

__global__ void __launch_bounds__(256) k_spmv_fp3(SpmvParams p)
{
    const fp3 k = fp_load(p.scale, 0);
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.rows; r += (size_t)gridDim.x * blockDim.x) {
        fp3 acc = fp_zero();                                  // sum of data x data products: scale 2^181
        const uint64_t e = p.row_ptr[r + 1];
        for (uint64_t t = p.row_ptr[r]; t < e; ++t) acc = fp_add(acc, fp_mul(fp_load(p.vec, p.col[t]), fp_load(p.coeff, t)));
        fp3 v = fp_mul(acc, k);
        if (p.accumulate) v = fp_add(v, fp_load(p.out, r));
        fp_store(p.out, r, v);
    }
}

// ---- one factor (1 + u^(2^k)) of the power-series inverse ------------------------------------------------------------------
#define PDIV_MAX_TERMS 66
struct PolyDivParams {
    const uint64_t *src;
    uint64_t *dst;
    uint64_t consts[3 * PDIV_MAX_TERMS];     // nterms elements (fp3: 2^203 form), in the argument block: no upload launch per pass
    size_t off[PDIV_MAX_TERMS];
    size_t M;
    int nterms;
};

__global__ void __launch_bounds__(256) k_polydiv_pass_gf192(PolyDivParams p)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.M; j += (size_t)gridDim.x * blockDim.x) {
        gf192 acc = gf_load(p.src, j);
        for (int t = 0; t < p.nterms; ++t) {
            const size_t s = j + p.off[t];
            if (s < p.M) gf_add_to(acc, gf_mul_uniform(gf_load(p.src, s), gf_load(p.consts, t)));
        }
        gf_store(p.dst, j, acc);
    }
}

__global__ void __launch_bounds__(256) k_polydiv_pass_fp3(PolyDivParams p)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.M; j += (size_t)gridDim.x * blockDim.x) {
        fp3 acc = fp_load(p.src, j);
        for (int t = 0; t < p.nterms; ++t) {
            const size_t s = j + p.off[t];
            if (s < p.M) acc = fp_add(acc, fp_mul(fp_load(p.src, s), fp_load(p.consts, t)));
        }
        fp_store(p.dst, j, acc);
    }
}

// raw-word constants of the F_p kernels: 2^214, 2^192, p - 2
static void fp_consts(uint64_t (&c)[9])
{
    const hfp3 k192 = hfp3::one(), k214 = k192.table_form().table_form();
    memcpy(c, k214.w, 24); memcpy(c + 3, k192.w, 24);
    c[6] = hfp3::P[0] - 2; c[7] = hfp3::P[1]; c[8] = hfp3::P[2];
}

struct Term { size_t off; uint64_t c[3]; };

// runs the passes; `consts_of_pass(k)` yields the terms of the factor 1 + u^(2^k) (offsets already scaled)
template<typename PassTerms, typename Launch>
static int run_division(const uint64_t *d_high, size_t M, size_t min_offset, uint64_t *d_quotient, PassTerms terms_of_pass, Launch launch)
{
    int passes = 0;
    for (size_t o = min_offset; o != 0 && o < M; o <<= 1) ++passes;         // offsets stay below 2 M <= 2^41: no overflow
    if (passes == 0) {
        if (d_high != d_quotient) { const int crc_ = iopx::copy_d2d(d_quotient, d_high, M * 24); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }
    TmpBuf ping;
    int rc;
    if (passes > 1 && (rc = ping.alloc(M * 24)) != IOPX_OK) return rc;
    const uint64_t *src = d_high;
    for (int k = 0; k < passes; ++k) {
        // the last pass writes the quotient; before that alternate between the temporary and the quotient buffer
        uint64_t *dst = ((passes - 1 - k) & 1) ? ping.u64() : d_quotient;
        const std::vector<Term> terms = terms_of_pass(k);
        PolyDivParams p;
        memset(&p, 0, sizeof(p));
        for (const Term &t : terms) {
            if (t.off >= M) continue;
            memcpy(p.consts + 3 * p.nterms, t.c, 24);
            p.off[p.nterms++] = t.off;
        }
        if (p.nterms == 0) {        // nothing reaches back into the quotient: the factor is 1 on this range
            if (src != dst) { const int crc_ = iopx::copy_d2d(dst, src, M * 24); if (crc_ != IOPX_OK) return crc_; }
            src = dst;
            continue;
        }
        p.src = src; p.dst = dst; p.M = M;
        if ((rc = launch(p)) != IOPX_OK) return rc;
        src = dst;
    }
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_gf192_add_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    if (!d_a || !d_b || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    { ProfScope ps_("k_gf192_add"); hipLaunchKernelGGL(k_gf192_add, dim3(eo_grid(count)), dim3(256), 0, stream(), d_out, d_a, d_b, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_gf192_inv_dev(const uint64_t *d_a, uint64_t *d_out, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    if (!d_a || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    { ProfScope ps_("k_gf192_inv"); hipLaunchKernelGGL(k_gf192_inv, dim3(eo_grid(count)), dim3(256), 0, stream(), d_out, d_a, count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

static int fp3_elementwise(int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    if (!d_a || (op != 2 && !d_b) || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    uint64_t c[9];
    fp_consts(c);
    TmpBuf dc;
    if ((rc = dc.alloc(sizeof(c))) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, c, sizeof(c))) != IOPX_OK) return rc;
    if (op == 0) { ProfScope ps_("k_fp3_mul"); hipLaunchKernelGGL(k_fp3_mul, dim3(eo_grid(count)), dim3(256), 0, stream(), d_out, d_a, d_b, (const uint64_t *)dc.u64(), count); }
    else if (op == 1) { ProfScope ps_("k_fp3_sub"); hipLaunchKernelGGL(k_fp3_sub, dim3(eo_grid(count)), dim3(256), 0, stream(), d_out, d_a, d_b, count); }
    else { ProfScope ps_("k_fp3_inv"); hipLaunchKernelGGL(k_fp3_inv, dim3(eo_grid(count)), dim3(256), 0, stream(), d_out, d_a, (const uint64_t *)dc.u64(), count); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_fp3_mul_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count) { return fp3_elementwise(0, d_a, d_b, d_out, count); }
int iopx_fp3_sub_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count) { return fp3_elementwise(1, d_a, d_b, d_out, count); }
int iopx_fp3_inv_dev(const uint64_t *d_a, uint64_t *d_out, size_t count) { return fp3_elementwise(2, d_a, nullptr, d_out, count); }

static int lincomb_common(const void *const *d_oracles, size_t num, const uint64_t *coeffs, const uint64_t *constant, size_t n, uint64_t *d_out,
                          bool prime_field)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_oracles || !coeffs || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (num == 0 || num > LINCOMB_MAX) return fail(IOPX_ERR_INVALID_ARGUMENT, "Random Linear Combination Oracle: Expected same number of evaluations as in registration.");
    std::vector<uint64_t> hc(coeffs, coeffs + 3 * num);
    if (prime_field) for (size_t i = 0; i < num; ++i) { const hfp3 t = hfp3::from_words(coeffs + 3 * i).table_form(); memcpy(&hc[3 * i], t.w, 24); }
    if (constant) hc.insert(hc.end(), constant, constant + 3);
    TmpBuf dc;
    if ((rc = dc.alloc(hc.size() * 8)) != IOPX_OK) return rc;
    if ((rc = upload(dc.p, hc.data(), hc.size() * 8)) != IOPX_OK) return rc;
    LincombParams p;
    memset(&p, 0, sizeof(p));
    for (size_t i = 0; i < num; ++i) { if (!d_oracles[i]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null oracle"); p.o[i] = (const uint64_t *)d_oracles[i]; }
    p.c = dc.u64(); p.out = d_out; p.num = (int)num; p.n = n; p.has_constant = constant ? 1 : 0;
    if (prime_field) { ProfScope ps_("k_lincomb_fp3", (num + 1) * n * 24); hipLaunchKernelGGL(k_lincomb_fp3, dim3(eo_grid(n)), dim3(256), 0, stream(), p); }
    else { ProfScope ps_("k_lincomb_gf192", (num + 1) * n * 24); hipLaunchKernelGGL(k_lincomb_gf192, dim3(eo_grid(n)), dim3(256), 0, stream(), p); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_lincomb_gf192_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, size_t n, uint64_t *d_out)
{
    return lincomb_common(d_oracles, num_oracles, coefficients, nullptr, n, d_out, false);
}
int iopx_lincomb_fp3_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, size_t n, uint64_t *d_out)
{
    return lincomb_common(d_oracles, num_oracles, coefficients, nullptr, n, d_out, true);
}
// sum_i c_i o_i + constant: single_matrix_denominator::evaluated_contents (libiop/protocols/encoded/lincheck/holographic_lincheck_aux.tcc:117-143)
int iopx_lincomb_affine_gf192_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, const uint64_t *constant, size_t n,
                                  uint64_t *d_out)
{
    if (!constant) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    return lincomb_common(d_oracles, num_oracles, coefficients, constant, n, d_out, false);
}
int iopx_lincomb_affine_fp3_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, const uint64_t *constant, size_t n,
                                uint64_t *d_out)
{
    if (!constant) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    return lincomb_common(d_oracles, num_oracles, coefficients, constant, n, d_out, true);
}

static int spmv_common(const uint64_t *d_row_ptr, const uint32_t *d_col, const uint64_t *d_coeff, size_t rows, const uint64_t *d_vec,
                       const uint64_t *scale, int accumulate, uint64_t *d_out, bool prime_field)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (rows == 0) return IOPX_OK;
    if (!d_row_ptr || !d_col || !d_coeff || !d_vec || !d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    SpmvParams p;
    p.row_ptr = d_row_ptr; p.col = d_col; p.coeff = d_coeff; p.vec = d_vec; p.out = d_out; p.rows = rows; p.accumulate = accumulate; p.scale = nullptr;
    TmpBuf ds;
    if (prime_field) {
        const hfp3 k = (scale ? hfp3::from_words(scale) : hfp3::one()).table_form().table_form();     // r 2^214 raw
        if ((rc = ds.alloc(24)) != IOPX_OK) return rc;
        if ((rc = upload(ds.p, k.w, 24)) != IOPX_OK) return rc;
        p.scale = ds.u64();
        { ProfScope ps_("k_spmv_fp3"); hipLaunchKernelGGL(k_spmv_fp3, dim3(eo_grid(rows)), dim3(256), 0, stream(), p); }
    } else {
        if (scale) {
            if ((rc = ds.alloc(24)) != IOPX_OK) return rc;
            if ((rc = upload(ds.p, scale, 24)) != IOPX_OK) return rc;
            p.scale = ds.u64();
        }
        { ProfScope ps_("k_spmv_gf192"); hipLaunchKernelGGL(k_spmv_gf192, dim3(eo_grid(rows)), dim3(256), 0, stream(), p); }
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_spmv_gf192_dev(const uint64_t *d_row_ptr, const uint32_t *d_col, const uint64_t *d_coeff, size_t rows, const uint64_t *d_vec,
                        const uint64_t *scale, int accumulate, uint64_t *d_out)
{
    return spmv_common(d_row_ptr, d_col, d_coeff, rows, d_vec, scale, accumulate, d_out, false);
}
int iopx_spmv_fp3_dev(const uint64_t *d_row_ptr, const uint32_t *d_col, const uint64_t *d_coeff, size_t rows, const uint64_t *d_vec,
                      const uint64_t *scale, int accumulate, uint64_t *d_out)
{
    return spmv_common(d_row_ptr, d_col, d_coeff, rows, d_vec, scale, accumulate, d_out, true);
}

int iopx_poly_div_vanishing_gf192_dev(const uint64_t *d_poly, size_t n_coeffs, const uint64_t *basis, size_t dim, const uint64_t *shift,
                                      uint64_t *d_quotient)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_poly || (dim > 0 && !basis) || !shift || !d_quotient) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (dim > 62) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension too large");
    const size_t N = (size_t)1 << dim;
    if (n_coeffs <= N) return IOPX_OK;                       // empty quotient (linearized_polynomial.tcc:247-255)
    const size_t M = n_coeffs - N;
    // Z = sum_{i <= dim} lin[i] X^(2^i) + lin(shift): the subspace polynomial built factor by factor (vanishing_polynomial.tcc:373-395)
    std::vector<hgf192> lin(1, hgf192::one());
    auto eval = [&](const hgf192 &x) { hgf192 r = hgf192::zero(), xp = x; for (size_t i = 0; i < lin.size(); ++i) { r += lin[i] * xp; xp = xp.squared(); } return r; };
    for (size_t k = 0; k < dim; ++k) {
        const hgf192 zb = eval(hgf192::from_words(basis + 3 * k));
        std::vector<hgf192> nxt(lin.size() + 1, hgf192::zero());
        for (size_t i = 0; i < lin.size(); ++i) { nxt[i + 1] += lin[i].squared(); nxt[i] += lin[i] * zb; }
        lin.swap(nxt);
    }
    if (!(lin[dim] == hgf192::one())) return fail(IOPX_ERR_LOGIC, "vanishing polynomial is not monic");
    const hgf192 c_const = eval(hgf192::from_words(shift));
    std::vector<hgf192> cur;                                 // c_i^(2^k), squared once per pass
    std::vector<size_t> base_off;
    for (size_t i = 0; i < dim; ++i) if (!lin[i].is_zero()) { cur.push_back(lin[i]); base_off.push_back(N - ((size_t)1 << i)); }
    if (!c_const.is_zero()) { cur.push_back(c_const); base_off.push_back(N); }
    if (cur.size() > PDIV_MAX_TERMS) return fail(IOPX_ERR_INVALID_ARGUMENT, "too many terms");
    size_t min_off = 0;
    for (size_t o : base_off) if (min_off == 0 || o < min_off) min_off = o;
    int done = 0;
    auto terms_of_pass = [&](int k) {
        while (done < k) { for (hgf192 &c : cur) c = c.squared(); ++done; }
        std::vector<Term> t;
        for (size_t i = 0; i < cur.size(); ++i) {
            Term x;
            x.off = base_off[i] << k;
            memcpy(x.c, cur[i].w, 24);
            t.push_back(x);
        }
        return t;
    };
    auto launch = [&](const PolyDivParams &p) -> int {
        { ProfScope ps_("k_polydiv_pass_gf192", 2 * (size_t)p.M * 24); hipLaunchKernelGGL(k_polydiv_pass_gf192, dim3(eo_grid(p.M)), dim3(256), 0, stream(), p); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    };
    if (cur.empty()) min_off = 0;
    return run_division(d_poly + 3 * N, M, min_off, d_quotient, terms_of_pass, launch);
}

int iopx_poly_div_vanishing_fp3_dev(const uint64_t *d_poly, size_t n_coeffs, size_t log_order, const uint64_t *shift, uint64_t *d_quotient)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_poly || !shift || !d_quotient) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (log_order > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_order %zu exceeds the 2-adicity of the field", log_order);
    const size_t N = (size_t)1 << log_order;
    if (n_coeffs <= N) return IOPX_OK;
    const size_t M = n_coeffs - N;
    // Z = X^N - c, c = shift^N (vanishing_polynomial.tcc:14-25): Q_j = P_{j+N} + c Q_{j+N}
    hfp3 cur = hfp3::from_words(shift).pow((uint64_t)N);
    int done = 0;
    auto terms_of_pass = [&](int k) {
        while (done < k) { cur = cur.squared(); ++done; }
        Term x;
        x.off = N << k;
        const hfp3 t = cur.table_form();
        memcpy(x.c, t.w, 24);
        return std::vector<Term>(1, x);
    };
    auto launch = [&](const PolyDivParams &p) -> int {
        { ProfScope ps_("k_polydiv_pass_fp3"); hipLaunchKernelGGL(k_polydiv_pass_fp3, dim3(eo_grid(p.M)), dim3(256), 0, stream(), p); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    };
    return run_division(d_poly + 3 * N, M, N, d_quotient, terms_of_pass, launch);
}

} // extern "C"
