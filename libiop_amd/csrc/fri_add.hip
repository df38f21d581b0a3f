// FRI prover fold over GF(2^192) for affine-subspace domains on gfx950.
//
// Replaces additive_evaluate_next_f_i_over_entire_domain (libiop/protocols/ldt/fri/fri_aux.tcc:36-103):
// next[j] = P_j(x_i), P_j the interpolant of f_i on the j-th contiguous coset of size 2^eta.
// The reference computes P_j(x_i) with one field inversion per coset (:92-99).  The value is unique, so
// this kernel uses the inversion-free nested form instead: a coset of size 2^eta is folded eta times
// by two,
//     g[J] = f[2J] + (f[2J] + f[2J+1]) * (x + v_{2J}) / b_0 ,
// each time over the domain derived by q(X) = X^2 + b_0 X (basis q(b_1..), shift q(s), point q(x)).
// (x + v_{2J}) / b_0 is GF(2)-affine in the bits of J, so the per-pair multiplier is a subset sum of
// m-1 host-prepared constants.  When x lies in the coset the formula returns f at x, which is what
// the reference's special case (:77-86) returns.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gf192_dev.h"
#include "gf192_host.h"
#include "runtime.h"

namespace iopx {

struct FoldParams {
    const uint64_t *src;    // 2 * n_out elements
    uint64_t *dst;          // n_out elements
    const uint64_t *consts; // [0] = (x + s) / b0, [1 + k] = b_{k+1} / b0, k < nbits
    int nbits;              // log2(n_out)
    size_t n_out;
};

__global__ void k_fri_fold2(FoldParams p)
{
    for (size_t J = (size_t)blockIdx.x * blockDim.x + threadIdx.x; J < p.n_out; J += (size_t)gridDim.x * blockDim.x) {
        gf192 mult = gf_load(p.consts, 0);
        for (int k = 0; k < p.nbits; ++k) {
            if ((J >> k) & 1) gf_add_to(mult, gf_load(p.consts, 1 + k));
        }
        const gf192 f0 = gf_load(p.src, 2 * J), f1 = gf_load(p.src, 2 * J + 1);
        gf192 r = gf_mul(gf_add(f0, f1), mult);
        gf_add_to(r, f0);
        gf_store(p.dst, J, r);
    }
}

// One kernel per FRI round for cosets of 2^eta elements, eta <= 3: a lane reads its whole coset (2^eta consecutive elements),
// folds it eta times in registers and writes one element, so a round moves (1 + 2^-eta) n elements instead of the
// (1 + 2 (1/2 + ... )) n of eta chained launches.  The level-e multiplier of pair J_e = C 2^(eta-1-e) + p of coset C is
//     A_e + sum_k bit_k(J_e) D_e[k] = [A_e + sum_k' bit_k'(C) D_e[eta-1-e+k']] + sum_{k < eta-1-e} bit_k(p) D_e[k] :
// the bracket splits into bits 0-5 of C — the lane id when a wave owns 64 consecutive cosets: computed once per kernel — and the
// wave-uniform bits >= 6 (uniform branches); the tail depends on the in-coset pair p only.
#define FOLD_MAX_ETA 3
struct FusedFoldParams {
    const uint64_t *src;
    uint64_t *dst;
    const uint64_t *consts[FOLD_MAX_ETA];   // level e: [0] = A_e, [1 + k] = D_e[k]
    int nbits[FOLD_MAX_ETA];                // number of D_e entries
    int eta;
    size_t n_out;
};

template<int ETA>
__global__ void __launch_bounds__(256) k_fri_fold_fused(FusedFoldParams p)
{
    // per-lane part of every level's multiplier: bits 0-5 of the coset index — the lane id within the wave for every iteration
    // of the grid-stride loop (workgroups are multiples of 64 lanes), so it is computed once
    gf192 lane_part[ETA];
    uint32_t have_low = 0xffffffffu;
    for (size_t base = (size_t)blockIdx.x * blockDim.x; base < p.n_out; base += (size_t)gridDim.x * blockDim.x) {
        const size_t C = base + threadIdx.x;
        const uint32_t low = (uint32_t)(C & 63);
        if (low != have_low) {
#pragma unroll
            for (int e = 0; e < ETA; ++e) {
                const int skip = ETA - 1 - e;           // D_e entries owned by the in-coset pair index
                gf192 t = gf_zero();
                for (int k = 0; k < 6 && skip + k < p.nbits[e]; ++k) if ((low >> k) & 1) gf_add_to(t, gf_load(p.consts[e], 1 + skip + k));
                lane_part[e] = t;
            }
            have_low = low;
        }
        const uint32_t chi = __builtin_amdgcn_readfirstlane((uint32_t)(C >> 6));       // bits >= 6: wave-uniform
        gf192 v[1 << ETA];
        if (C < p.n_out) {
#pragma unroll
            for (int i = 0; i < (1 << ETA); ++i) v[i] = gf_load(p.src, (C << ETA) + i);
        }
#pragma unroll
        for (int e = 0; e < ETA; ++e) {
            const int skip = ETA - 1 - e;
            gf192 m = gf_add(gf_load(p.consts[e], 0), lane_part[e]);
            for (int k = 6; skip + k < p.nbits[e]; ++k) if ((chi >> (k - 6)) & 1) gf_add_to(m, gf_load(p.consts[e], 1 + skip + k));
            if (C < p.n_out) {
#pragma unroll
                for (int q = 0; q < (1 << skip); ++q) {
                    gf192 mq = m;
#pragma unroll
                    for (int k = 0; k < skip; ++k) if ((q >> k) & 1) gf_add_to(mq, gf_load(p.consts[e], 1 + k));
                    gf192 r = gf_mul(gf_add(v[2 * q], v[2 * q + 1]), mq);
                    gf_add_to(r, v[2 * q]);
                    v[q] = r;
                }
            }
        }
        if (C < p.n_out) gf_store(p.dst, C, v[0]);
    }
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_fri_fold_add_gf192_dev(const uint64_t *d_f_i, const uint64_t *basis, size_t m, const uint64_t *shift,
                                size_t coset_size, const uint64_t *x_i, uint64_t *d_next)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    if (!d_f_i || !d_next || !shift || !x_i || (m && !basis)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (coset_size == 0 || (coset_size & (coset_size - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "coset size %zu is not a power of two", coset_size);
    const int eta = (int)ceil_log2(coset_size);
    if ((size_t)eta > m) return fail(IOPX_ERR_INVALID_ARGUMENT, "coset size %zu exceeds the domain size", coset_size);
    const size_t n = (size_t)1 << m;
    if (eta == 0) {     // cosets of one element: the interpolant is the constant f(v)
        { const int crc_ = iopx::copy_d2d(d_next, d_f_i, n * 24); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }

    std::vector<hgf192> b(m);
    for (size_t i = 0; i < m; ++i) b[i] = hgf192::from_words(basis + 3 * i);
    hgf192 s = hgf192::from_words(shift), x = hgf192::from_words(x_i);

    // constants of all eta levels, uploaded once
    std::vector<uint64_t> hc;
    std::vector<size_t> off(eta);
    for (int e = 0; e < eta; ++e) {
        if (b[0].is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "FRI fold: basis vectors are linearly dependent");
        const hgf192 b0 = b[0], b0inv = b0.inverse();
        off[e] = hc.size() / 3;
        const hgf192 a = (x + s) * b0inv;
        hc.insert(hc.end(), a.w, a.w + 3);
        for (size_t k = 1; k < b.size(); ++k) {
            const hgf192 dk = b[k] * b0inv;
            hc.insert(hc.end(), dk.w, dk.w + 3);
        }
        // derived domain: q(X) = X^2 + b0 X
        std::vector<hgf192> nb;
        for (size_t k = 1; k < b.size(); ++k) nb.push_back(b[k].squared() + b0 * b[k]);
        s = s.squared() + b0 * s;
        x = x.squared() + b0 * x;
        b.swap(nb);
    }
    TmpBuf dc;
    if ((rc = dc.alloc(hc.size() * 8)) != IOPX_OK) return rc;
    { int urc_ = upload(dc.p, hc.data(), hc.size() * 8); if (urc_ != IOPX_OK) return urc_; }

    if (eta <= FOLD_MAX_ETA) {
        FusedFoldParams fp;
        memset(&fp, 0, sizeof(fp));
        fp.src = d_f_i; fp.dst = d_next; fp.eta = eta; fp.n_out = n >> eta;
        for (int e = 0; e < eta; ++e) { fp.consts[e] = dc.u64() + 3 * off[e]; fp.nbits[e] = (int)m - 1 - e; }
        size_t grid = (fp.n_out + 255) / 256;
        if (grid > 16384) grid = 16384;
        if (grid < 1) grid = 1;
        const size_t bytes = (n + fp.n_out) * 24;
        if (eta == 1) { ProfScope ps_("k_fri_fold_fused_eta1", bytes); hipLaunchKernelGGL(k_fri_fold_fused<1>, dim3((unsigned)grid), dim3(256), 0, stream(), fp); }
        else if (eta == 2) { ProfScope ps_("k_fri_fold_fused_eta2", bytes); hipLaunchKernelGGL(k_fri_fold_fused<2>, dim3((unsigned)grid), dim3(256), 0, stream(), fp); }
        else { ProfScope ps_("k_fri_fold_fused_eta3", bytes); hipLaunchKernelGGL(k_fri_fold_fused<3>, dim3((unsigned)grid), dim3(256), 0, stream(), fp); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    }
    TmpBuf tmp[2];
    const uint64_t *src = d_f_i;
    size_t cur = n;
    for (int e = 0; e < eta; ++e) {
        const size_t n_out = cur >> 1;
        uint64_t *dst = d_next;
        if (e != eta - 1) {
            if ((rc = tmp[e & 1].alloc(n_out * 24)) != IOPX_OK) return rc;
            dst = tmp[e & 1].u64();
        }
        FoldParams p;
        p.src = src; p.dst = dst; p.consts = dc.u64() + 3 * off[e];
        p.nbits = (int)m - 1 - e; p.n_out = n_out;
        size_t grid = (n_out + 255) / 256;
        if (grid > 16384) grid = 16384;
        if (grid < 1) grid = 1;
        { ProfScope ps_("k_fri_fold2"); hipLaunchKernelGGL(k_fri_fold2, dim3((unsigned)grid), dim3(256), 0, stream(), p); }
        src = dst;
        cur = n_out;
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;                                 // constants / temporaries are released in stream order
}

int iopx_fri_fold_add_gf192(const uint64_t *f_i, const uint64_t *basis, size_t m, const uint64_t *shift,
                            size_t coset_size, const uint64_t *x_i, uint64_t *next)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (m > 40) return fail(IOPX_ERR_INVALID_ARGUMENT, "domain dimension %zu too large", m);
    if (coset_size == 0 || (coset_size & (coset_size - 1)) || coset_size > ((size_t)1 << m))
        return fail(IOPX_ERR_INVALID_ARGUMENT, "bad coset size %zu", coset_size);
    const size_t n = (size_t)1 << m, n_out = n / coset_size;
    DevBuf din, dout;
    if ((rc = din.alloc(n * 24)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(n_out * 24)) != IOPX_OK) return rc;
    IOPX_HIP(copy_h2d(din.p, f_i, n * 24, stream()));
    rc = iopx_fri_fold_add_gf192_dev(din.u64(), basis, m, shift, coset_size, x_i, dout.u64());
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(next, dout.p, n_out * 24, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

// FRI_protocol::compute_domains, additive branch (libiop/protocols/ldt/fri/fri_ldt.tcc:310-338): L^(i+1) has basis q(basis[eta_i..])
// and shift q(shift), q = the subspace polynomial of span(basis[0..eta_i)) (localizer_polynomial.tcc:3-27,
// vanishing_polynomial.tcc:373-395).  Host-only metadata: out_bases receives the bases of L^(1), L^(2), ... back to back
// (sum_i dim(L^(i)) elements), out_shifts one element per derived domain.
int iopx_fri_domains_gf192(const uint64_t *basis, size_t m, const uint64_t *shift, const size_t *localization, size_t num_reductions,
                           uint64_t *out_bases, uint64_t *out_shifts)
{
    if ((m > 0 && !basis) || !shift || (num_reductions > 0 && (!localization || !out_bases || !out_shifts))) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<hgf192> b(m);
    for (size_t i = 0; i < m; ++i) b[i] = hgf192::from_words(basis + 3 * i);
    hgf192 s = hgf192::from_words(shift);
    size_t off = 0;
    for (size_t r = 0; r < num_reductions; ++r) {
        const size_t eta = localization[r];
        if (eta > b.size()) return fail(IOPX_ERR_INVALID_ARGUMENT, "localization parameters exceed the domain dimension");
        std::vector<hgf192> q(1, hgf192::one());                 // coefficient i multiplies X^(2^i)
        auto eval = [&](const hgf192 &x) { hgf192 v = hgf192::zero(), xp = x; for (size_t i = 0; i < q.size(); ++i) { v += q[i] * xp; xp = xp.squared(); } return v; };
        for (size_t k = 0; k < eta; ++k) {
            const hgf192 qb = eval(b[k]);
            std::vector<hgf192> nxt(q.size() + 1, hgf192::zero());
            for (size_t i = 0; i < q.size(); ++i) { nxt[i + 1] += q[i].squared(); nxt[i] += q[i] * qb; }
            q.swap(nxt);
        }
        std::vector<hgf192> nb;
        for (size_t k = eta; k < b.size(); ++k) nb.push_back(eval(b[k]));
        s = eval(s);
        b.swap(nb);
        for (const hgf192 &v : b) { memcpy(out_bases + 3 * off, v.w, 24); ++off; }
        memcpy(out_shifts + 3 * r, s.w, 24);
    }
    return IOPX_OK;
}

} // extern "C"
