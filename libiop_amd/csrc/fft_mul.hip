// Multiplicative-coset FFT / IFFT and FRI fold over the 181-bit prime field (libff edwards_Fr) for gfx950.
//
// Replaces, for multiplicative domains (reference paths relative to /root/reference):
//   multiplicative_FFT_degree_aware      libiop/algebra/fft.tcc:236-317   (a[i] = P(shift * g^i), natural order)
//   multiplicative_IFFT_internal         libiop/algebra/fft.tcc:343-361   -> libfqfft basic_radix2_domain::iFFT / icosetFFT
//   IFFT_of_known_degree (mult.)         libiop/algebra/fft.tcc:435-456   (strided gather + IFFT on the sub-coset)
//   multiplicative_evaluate_next_f_i_... libiop/protocols/ldt/fri/fri_aux.tcc:106-249
//
// Forward transform = the reference's structure: coefficients (pre-scaled by shift^k, fft.tcc:246-249) are read in
// bit-reversed order, each value is replicated over the 2^(log n - ceil(log2 len)) low index bits (:263-289), and
// only the last ceil(log2 len) radix-2 levels run (:293-315), in LDS tiles of up to 12 index bits per HBM sweep,
// with the reference's twiddle cache layout (subgroup.tcc:117-144) kept in HBM per domain.
// Inverse = radix-2 FFT with g^-1 on the bit-reversed evaluations, scaled by n^-1 (and shift^-i for a coset) in the
// last pass.  The fold uses the inversion-free nested form: a coset {j + k n/c} is folded log2(c) times by two,
//     g[j] = ((a + b) + (a - b) * x / (shift * g^j)) / 2,   a = f[j], b = f[j + n/2],
// each time over the squared domain (shift^2, g^2, x^2) — the unique interpolant value the reference computes with
// one global batch inversion (:230-231).
#include <hip/hip_runtime.h>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "fp3_dev.h"
#include "fp3_host.h"
#include "runtime.h"

namespace iopx {

// tile geometry of k_mfft_pass: 2048-element tiles, 16 contiguous columns in the strided passes
static const int MF_TILE_BITS = 11;
static const int MF_COLS = 4;

__device__ __forceinline__ fp3 mlds_get(const uint64_t *s, int E, int li)
{
    const uint64_t a = s[li], b = s[E + li], c = s[2 * E + li];
    fp3 r;
    r.w[0] = (uint32_t)a; r.w[1] = (uint32_t)(a >> 32);
    r.w[2] = (uint32_t)b; r.w[3] = (uint32_t)(b >> 32);
    r.w[4] = (uint32_t)c; r.w[5] = (uint32_t)(c >> 32);
    return r;
}

__device__ __forceinline__ void mlds_put(uint64_t *s, int E, int li, const fp3 &v)
{
    s[li] = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32);
    s[E + li] = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32);
    s[2 * E + li] = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
}

// out[q] = init * prod_{k : bit k of q} sq[k]   (sq[k] = base^(2^k)), q < count
__global__ void k_fp_pow_direct(uint64_t *out, const uint64_t *sq, const uint64_t *init, int nbits, size_t count)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (size_t)gridDim.x * blockDim.x) {
        fp3 acc = fp_load(init, 0);
        for (int k = 0; k < nbits; ++k) {
            if ((q >> k) & 1) acc = fp_mul(acc, fp_load(sq, k));
        }
        fp_store(out, q, acc);
    }
}

// out[q] = out[q & 255] * hi[q >> 8]  for 256 <= q < count
__global__ void k_fp_pow_expand(uint64_t *out, const uint64_t *hi, size_t count)
{
    for (size_t q = 256 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (size_t)gridDim.x * blockDim.x) {
        fp_store(out, q, fp_mul(fp_load(out, q & 255), fp_load(hi, q >> 8)));
    }
}

// cache level with m = 2^b entries at offset m - 1: entry j = top[j << (logn - 1 - b)]   (subgroup.tcc:117-144)
__global__ void k_fp_cache_level(uint64_t *cache, int logn, int b)
{
    const size_t m = (size_t)1 << b;
    const uint64_t *top = cache + 3 * ((((size_t)1) << (logn - 1)) - 1);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (size_t)gridDim.x * blockDim.x) {
        fp_store(cache, m - 1 + j, fp_load(top, j << (logn - 1 - b)));
    }
}

// dst[k] = src[k] * hi[k >> 12] * lo[k & 4095]      (coset pre-scaling a[k] *= shift^k, fft.tcc:246-249)
__global__ void k_fp_scale_pow(uint64_t *dst, const uint64_t *src, const uint64_t *hi, const uint64_t *lo, size_t count)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (size_t)gridDim.x * blockDim.x) {
        fp_store(dst, k, fp_mul(fp_mul(fp_load(src, k), fp_load(hi, k >> 12)), fp_load(lo, k & 4095)));
    }
}

__global__ void k_fp_gather_stride(uint64_t *dst, const uint64_t *src, size_t stride, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * count; i += (size_t)gridDim.x * blockDim.x) {
        dst[i] = src[3 * ((i / 3) * stride) + (i % 3)];
    }
}

struct MfParams {
    const uint64_t *src;    // first pass: coefficient / evaluation array of n_src elements (gathered bit-reversed)
    uint64_t *dst;
    const uint64_t *cache;  // n - 1 twiddles, level b at offset 2^b - 1
    const uint64_t *sc_hi, *sc_lo;  // last pass: out[i] *= sc_hi[i >> 12] * sc_lo[i & 4095]   (null: no scaling)
    size_t n_src;
    int logn, logrho;       // index bits [logrho, logn) are active; the low logrho bits replicate
    int gather;
    int c, h, A;            // tile: columns on bits [0,c), rows on bits [h, h+A)
    int b_lo, b_hi;         // butterfly bits of this pass (ascending)
    int scale;              // 0 none, 1 sc_hi[0] only (n^-1), 2 two-level table
    int final;              // last pass of a transform: store canonical values (earlier passes store lazily reduced ones)
    // last pass: windows of the output written beside it — window w holds out[first + k << log_stride], k < 2^(logn - log_stride) — so that a caller
    // that needs the strided head of the codeword (the positions a known-degree interpolation reads, fft.tcc:435-456) does not sweep the codeword
    // again with 24 useful bytes per 128-byte line (k_gather_stride_words moved 3.2 x its useful bytes)
    // (scalar fields, not arrays: a dynamically indexed member would move the argument block out of SGPRs)
    uint64_t *win0_dst, *win1_dst;       // null: no such window
    uint32_t win0_first, win1_first;
    int win0_log_stride, win1_log_stride;
};

// R levels starting at global index bit b on 2^R elements per lane (local indices i0 | k << bl).  Level b + lev pairs k with
// k | 1 << lev; its twiddle index is the element's index below bit b + lev: lowidx plus the already-processed bits of k.
template<int R>
__device__ __forceinline__ void mfft_step(uint64_t *s, int E, const MfParams &p, size_t base, int cmask, int b, int tid, int nt)
{
    const int bl = b - p.h + p.c;                           // tile-local bit of level b
    for (int grp = tid; grp < (E >> R); grp += nt) {
        const int low = grp & ((1 << bl) - 1), high = grp >> bl;
        const int i0 = (high << (bl + R)) | low;
        const size_t gi0 = base | ((size_t)(i0 >> p.c) << p.h) | (size_t)(i0 & cmask);
        const size_t lowidx = gi0 & ((((size_t)1) << b) - 1);
        fp7 v[1 << R];
#pragma unroll
        for (int k = 0; k < (1 << R); ++k) v[k] = fp7_unpack(mlds_get(s, E, i0 | (k << bl)));
#pragma unroll
        for (int lev = 0; lev < R; ++lev) {
            const uint64_t *lvl = p.cache + 3 * ((((size_t)1) << (b + lev)) - 1);
#pragma unroll
            for (int q = 0; q < (1 << lev); ++q) {
                const fp7 w = fp7_unpack(fp_load(lvl, lowidx + ((size_t)q << b)));
#pragma unroll
                for (int k = 0; k < (1 << R); ++k) {
                    if ((k & ((1 << lev) - 1)) == q && !((k >> lev) & 1)) fp7_bfly(v[k], v[k | (1 << lev)], w);   // t = w * a[k+j+m] (fft.tcc:303-309)
                }
            }
        }
#pragma unroll
        for (int k = 0; k < (1 << R); ++k) mlds_put(s, E, i0 | (k << bl), fp7_pack(fp7_norm(v[k])));
    }
}

// WIN: the last pass of a transform that also writes windows of its output (a separate instantiation: the passes without windows keep their registers)
template<bool WIN>
__global__ void __launch_bounds__(512) k_mfft_pass(MfParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t iopx_smem[];
    uint64_t *s = iopx_smem;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int E = 1 << (p.c + p.A);
    const int midbits = p.h - p.c;
    const size_t o = blockIdx.x;
    const size_t mid = o & (((size_t)1 << midbits) - 1), hi = o >> midbits;
    const size_t base = (hi << (p.h + p.A)) | (mid << p.c);
    const int cmask = (1 << p.c) - 1;
    const int logd = p.logn - p.logrho;

    for (int li = tid; li < E; li += nt) {
        const size_t gi = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
        fp3 v;
        if (p.gather) {
            const size_t t = gi >> p.logrho;
            const size_t k = logd == 0 ? 0 : (size_t)(__brevll((unsigned long long)t) >> (64 - logd));
            v = k < p.n_src ? fp_load(p.src, k) : fp_zero();
        } else {
            v = fp_load(p.src, gi);
        }
        mlds_put(s, E, li, v);
    }
    __syncthreads();

    // radix-8 / 4 / 2 steps: a lane keeps 2^R elements in registers across R levels, so the tile makes one LDS round trip
    // and one barrier per three levels
    int b = p.b_lo;
    for (; b + 2 <= p.b_hi; b += 3) { mfft_step<3>(s, E, p, base, cmask, b, tid, nt); __syncthreads(); }
    if (b + 1 <= p.b_hi) { mfft_step<2>(s, E, p, base, cmask, b, tid, nt); __syncthreads(); b += 2; }
    if (b <= p.b_hi) { mfft_step<1>(s, E, p, base, cmask, b, tid, nt); __syncthreads(); }

    for (int li = tid; li < E; li += nt) {
        const size_t gi = base | ((size_t)(li >> p.c) << p.h) | (size_t)(li & cmask);
        fp3 v = mlds_get(s, E, li);                     // below 2^192, not necessarily canonical
        if (p.scale == 1) v = fp_mul(v, fp_load(p.sc_hi, 0));
        else if (p.scale == 2) v = fp_mul(fp_mul(v, fp_load(p.sc_hi, gi >> 12)), fp_load(p.sc_lo, gi & 4095));
        else if (p.final) v = fp7_canonical(fp7_unpack(v));
        fp_store(p.dst, gi, v);
        if (WIN) {
            if (p.win0_dst && (gi & ((((size_t)1) << p.win0_log_stride) - 1)) == p.win0_first) fp_store(p.win0_dst, gi >> p.win0_log_stride, v);
            if (p.win1_dst && (gi & ((((size_t)1) << p.win1_log_stride) - 1)) == p.win1_first) fp_store(p.win1_dst, gi >> p.win1_log_stride, v);
        }
    }
}

struct MfoldParams {
    const uint64_t *src;
    uint64_t *dst;
    const uint64_t *ginv;   // g^-j for j < n0/2 (top level of the inverse cache)
    const uint64_t *consts; // [0] = x / shift, [1] = 1/2    (of this level)
    size_t half;            // outputs = pairs (j, j + half)
    int stride_log;         // ginv index = j << stride_log
};

// g[j] = ((a + b) + (a - b) * (x / shift) * g^-j) / 2
__global__ void k_fri_fold2_mul(MfoldParams p)
{
    const fp3 xs = fp_load(p.consts, 0), inv2 = fp_load(p.consts, 1);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.half; j += (size_t)gridDim.x * blockDim.x) {
        const fp3 a = fp_load(p.src, j), b = fp_load(p.src, j + p.half);
        const fp3 c = fp_mul(xs, fp_load(p.ginv, j << p.stride_log));
        const fp3 r = fp_add(fp_add(a, b), fp_mul(fp_sub(a, b), c));
        fp_store(p.dst, j, fp_mul(r, inv2));
    }
}

// One kernel per FRI round for cosets of 2^ETA: output j (j < n / 2^ETA) needs f[j + t q], q = n / 2^ETA, t < 2^ETA (the
// strided coset of subgroup.tcc:175-197): a lane loads them, folds ETA times in registers and writes one element.  Level e
// works on an array of n >> e values: entry u pairs with u + (n >> (e + 1)), multiplier (x / shift)^(2^e) * g^-(u << e).
template<int ETA>
__global__ void __launch_bounds__(256) k_fri_fold_fused_mul(MfoldParams p)        // p.half = number of outputs q; p.consts: (xs_e, 1/2) per level
{
    const size_t q = p.half;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < q; j += (size_t)gridDim.x * blockDim.x) {
        fp3 v[1 << ETA];
#pragma unroll
        for (int t = 0; t < (1 << ETA); ++t) v[t] = fp_load(p.src, j + (size_t)t * q);
#pragma unroll
        for (int e = 0; e < ETA; ++e) {
            const fp3 xs = fp_load(p.consts, 2 * e), inv2 = fp_load(p.consts, 2 * e + 1);
            const int pairs = 1 << (ETA - 1 - e);
#pragma unroll
            for (int t = 0; t < pairs; ++t) {
                const size_t u = j + (size_t)t * q;                             // index in the level-e array
                const fp3 c = fp_mul(xs, fp_load(p.ginv, u << e));
                const fp3 a = v[t], b = v[t + pairs];
                v[t] = fp_mul(fp_add(fp_add(a, b), fp_mul(fp_sub(a, b), c)), inv2);
            }
        }
        fp_store(p.dst, j, v[0]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct MulPlan {
    int logn = 0;
    hfp3 g, ginv;
    DevBuf cache_fwd, cache_inv;        // n - 1 twiddles each, built on first use
    bool have_fwd = false, have_inv = false;
};

static std::mutex g_mplan_mu;
static std::map<std::vector<uint64_t>, std::unique_ptr<MulPlan>> g_mplans;

static void clear_pow_tables();

void clear_mul_plans()
{
    {
        std::lock_guard<std::mutex> lk(g_mplan_mu);
        g_mplans.clear();
    }
    clear_pow_tables();
}

static int mgrid(size_t work, int threads)
{
    size_t g = (work + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

// out[q] = init * base^q for q < 2^nb
static int fp_build_pow(uint64_t *out, const hfp3 &base, const hfp3 &init, int nb)
{
    std::vector<uint64_t> sq;
    hfp3 x = base;
    for (int k = 0; k < nb; ++k) { const hfp3 t = x.table_form(); sq.insert(sq.end(), t.w, t.w + 3); x = x.squared(); }
    const hfp3 init_t = init.table_form();
    const size_t init_at = sq.size();                       // one block: the squarings, then the initial value (one constant-carrying launch)
    sq.insert(sq.end(), init_t.w, init_t.w + 3);
    TmpBuf dsq;
    int rc;
    if ((rc = dsq.alloc(sq.size() * 8)) != IOPX_OK) return rc;
    { int urc_ = upload(dsq.p, sq.data(), sq.size() * 8); if (urc_ != IOPX_OK) return urc_; }
    struct { const uint64_t *p; const uint64_t *u64() const { return p; } } dinit = { dsq.u64() + init_at };
    const size_t count = (size_t)1 << nb;
    if (nb <= 14) {         // up to 14 products per entry: one launch beats the three of the expansion scheme for the small per-call tables
        { ProfScope ps_("k_fp_pow_direct"); hipLaunchKernelGGL(k_fp_pow_direct, dim3(mgrid(count, 256)), dim3(256), 0, stream(), out, (const uint64_t *)dsq.u64(), (const uint64_t *)dinit.u64(), nb, count); }
    } else {
        // out[0..256) = init * base^q ; hi[r] = (base^256)^r ; out[q] = out[q & 255] * hi[q >> 8]
        TmpBuf hi;
        if ((rc = hi.alloc((((size_t)1) << (nb - 8)) * 24)) != IOPX_OK) return rc;
        hfp3 b256 = base;
        for (int k = 0; k < 8; ++k) b256 = b256.squared();
        rc = fp_build_pow(hi.u64(), b256, hfp3::one(), nb - 8);
        if (rc != IOPX_OK) return rc;
        { ProfScope ps_("k_fp_pow_direct"); hipLaunchKernelGGL(k_fp_pow_direct, dim3(1), dim3(256), 0, stream(), out, (const uint64_t *)dsq.u64(), (const uint64_t *)dinit.u64(), 8, (size_t)256); }
        { ProfScope ps_("k_fp_pow_expand"); hipLaunchKernelGGL(k_fp_pow_expand, dim3(mgrid(count - 256, 256)), dim3(256), 0, stream(), out, (const uint64_t *)hi.u64(), count); }
    }
    return IOPX_OK;         // temporaries are released in stream order
}

static int build_cache(MulPlan &pl, bool inverse)
{
    ColdScope cold_("multiplicative FFT twiddle cache");
    const int logn = pl.logn;
    DevBuf &buf = inverse ? pl.cache_inv : pl.cache_fwd;
    const size_t n = (size_t)1 << logn;
    int rc = buf.alloc((n > 1 ? n - 1 : 1) * 24);
    if (rc != IOPX_OK) return rc;
    if (logn >= 1) {
        uint64_t *top = buf.u64() + 3 * ((n >> 1) - 1);
        rc = fp_build_pow(top, inverse ? pl.ginv : pl.g, hfp3::one(), logn - 1);
        if (rc != IOPX_OK) return rc;
        for (int b = 0; b < logn - 1; ++b) {
            { ProfScope ps_("k_fp_cache_level"); hipLaunchKernelGGL(k_fp_cache_level, dim3(mgrid((size_t)1 << b, 256)), dim3(256), 0, stream(), buf.u64(), logn, b); }
        }
        IOPX_HIP(hipStreamSynchronize(stream()));
    }
    (inverse ? pl.have_inv : pl.have_fwd) = true;
    return IOPX_OK;
}

static int get_mplan(int logn, const uint64_t *gen, MulPlan **out)
{
    std::vector<uint64_t> key(gen, gen + 3);
    key.push_back((uint64_t)logn);
    std::lock_guard<std::mutex> lk(g_mplan_mu);
    auto it = g_mplans.find(key);
    if (it != g_mplans.end()) { *out = it->second.get(); return IOPX_OK; }
    std::unique_ptr<MulPlan> pl(new MulPlan());
    pl->logn = logn;
    pl->g = hfp3::from_words(gen);
    if (pl->g.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "multiplicative FFT: zero generator");
    // the generator must have order exactly 2^logn
    if (!(pl->g.pow((uint64_t)1 << logn) == hfp3::one()) || (logn > 0 && pl->g.pow((uint64_t)1 << (logn - 1)) == hfp3::one()))
        return fail(IOPX_ERR_INVALID_ARGUMENT, "multiplicative FFT: generator does not have order 2^%d", logn);
    pl->ginv = pl->g.inverse();
    *out = pl.get();
    g_mplans[key] = std::move(pl);
    return IOPX_OK;
}

// two-level power tables: hi[q] = init * base^(4096 q) (q < 2^max(logc-12,0)), lo[r] = base^r (r < 4096)
struct TableKey {
    uint64_t w[7];
    bool operator<(const TableKey &o) const { return memcmp(w, o.w, sizeof(w)) < 0; }
};
static std::map<TableKey, std::shared_ptr<DevBuf>> g_pow_tables;        // device tables of build_two_level, dropped by clear_mul_plans()
static std::mutex g_pow_tables_mu;
static size_t pow_table_cap()
{
    static const size_t cap = (size_t)opt_range("IOPX_POW_TABLE_CAP", 256, 1, 1 << 20);   // the env override exists for the eviction test
    return cap;
}

static int cached_pow_table(const hfp3 &base, const hfp3 &init, int bits, bool use_cache, TmpBuf &out)
{
    const size_t bytes = (((size_t)1) << bits) * 24;
    int rc;
    if (!use_cache) {
        if ((rc = out.alloc(bytes)) != IOPX_OK) return rc;
        return fp_build_pow(out.u64(), base, init, bits);
    }
    TableKey key;
    memcpy(key.w, base.w, 24); memcpy(key.w + 3, init.w, 24); key.w[6] = (uint64_t)bits;
    std::lock_guard<std::mutex> lk(g_pow_tables_mu);
    auto it = g_pow_tables.find(key);
    if (it == g_pow_tables.end()) {
        // rare: drop the cache.  Tables borrowed by a call in progress stay alive through that call's TmpBuf (shared ownership); the
        // others are freed here — hipFree waits for the device, so kernels in flight that read them have finished by then
        if (g_pow_tables.size() >= pow_table_cap()) g_pow_tables.clear();
        std::shared_ptr<DevBuf> buf(new DevBuf());
        if ((rc = buf->alloc(bytes)) != IOPX_OK) return rc;
        if ((rc = fp_build_pow(buf->u64(), base, init, bits)) != IOPX_OK) return rc;
        it = g_pow_tables.emplace(key, std::move(buf)).first;
    }
    out.borrow(it->second->p, bytes, it->second);
    return IOPX_OK;
}

static void clear_pow_tables()
{
    std::lock_guard<std::mutex> lk(g_pow_tables_mu);
    g_pow_tables.clear();
}

int build_two_level(const hfp3 &base, const hfp3 &init, int logc, TmpBuf &hi, TmpBuf &lo, bool cache_hi)
{
    const int lo_bits = logc < 12 ? logc : 12, hi_bits = logc > 12 ? logc - 12 : 0;
    // lo[r] = base^r for r < 2^lo_bits (the kernels index it with j & 4095: entries past 2^lo_bits are never read when logc < 12)
    int rc = cached_pow_table(base, hfp3::one(), lo_bits, true, lo);
    if (rc != IOPX_OK) return rc;
    hfp3 b4096 = base;
    for (int k = 0; k < 12; ++k) b4096 = b4096.squared();
    return cached_pow_table(b4096, init, hi_bits, cache_hi, hi);
}

// runs the radix-2 levels on index bits [logrho, logn) (first pass gathers src bit-reversed), natural-order dst
struct MfWindows { int num = 0; uint64_t *dst[2] = { nullptr, nullptr }; uint32_t first[2] = { 0, 0 }; int log_stride[2] = { 0, 0 }; };

static int run_mfft(const uint64_t *cache, const uint64_t *src, size_t n_src, uint64_t *dst, int logn, int logrho,
                    int scale, const uint64_t *sc_hi, const uint64_t *sc_lo, const MfWindows *windows = nullptr)
{
    struct Pass { int c, h, A, b_lo, b_hi; };
    std::vector<Pass> passes;
    int b = logrho;
    if (logn <= MF_TILE_BITS) {
        passes.push_back({0, 0, logn, b, logn - 1});
        b = logn;
    } else if (logrho < MF_TILE_BITS) {
        passes.push_back({0, 0, MF_TILE_BITS, b, MF_TILE_BITS - 1});
        b = MF_TILE_BITS;
    }
    while (b < logn) {
        int A = MF_TILE_BITS - MF_COLS;
        if (b + A > logn) A = logn - b;
        int c = MF_TILE_BITS - A;
        if (c > b) c = b;
        passes.push_back({c, b, A, b, b + A - 1});
        b += A;
    }
    if (passes.empty()) passes.push_back({0, 0, logn < MF_TILE_BITS ? logn : MF_TILE_BITS, 1, 0});     // replication only
    for (size_t i = 0; i < passes.size(); ++i) {
        const Pass &ps = passes[i];
        MfParams p;
        memset(&p, 0, sizeof(p));
        p.src = i == 0 ? src : dst;
        p.dst = dst;
        p.cache = cache;
        p.n_src = n_src;
        p.logn = logn; p.logrho = logrho;
        p.gather = (i == 0);
        p.c = ps.c; p.h = ps.h; p.A = ps.A; p.b_lo = ps.b_lo; p.b_hi = ps.b_hi;
        if (i + 1 == passes.size()) {
            p.scale = scale; p.sc_hi = sc_hi; p.sc_lo = sc_lo; p.final = 1;
            if (windows && windows->num > 0) { p.win0_dst = windows->dst[0]; p.win0_first = windows->first[0]; p.win0_log_stride = windows->log_stride[0]; }
            if (windows && windows->num > 1) { p.win1_dst = windows->dst[1]; p.win1_first = windows->first[1]; p.win1_log_stride = windows->log_stride[1]; }
        }
        const int tbits = ps.c + ps.A;
        const size_t lds = ((size_t)24) << tbits;
        const size_t blocks = (size_t)1 << (logn - tbits);
        const int threads = (1 << tbits) >= 512 ? (1 << tbits) / 8 : 64;          // one radix-8 group per lane and step
        const bool win = p.win0_dst != nullptr;
        if (lds > 64 * 1024) IOPX_HIP(hipFuncSetAttribute(win ? (const void *)k_mfft_pass<true> : (const void *)k_mfft_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // algorithmic bytes: the pass reads and writes the 2^logn-element vector once; products: one per radix-2 butterfly of its levels
        { ProfScope ps_("k_mfft_pass", ((size_t)48) << logn, (((size_t)1 << logn) >> 1) * (size_t)(ps.b_hi >= ps.b_lo ? ps.b_hi - ps.b_lo + 1 : 0));
          if (win) hipLaunchKernelGGL(k_mfft_pass<true>, dim3((unsigned)blocks), dim3(threads), lds, stream(), p);
          else hipLaunchKernelGGL(k_mfft_pass<false>, dim3((unsigned)blocks), dim3(threads), lds, stream(), p); }
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_mul_fft_fp3_dev(const uint64_t *d_coeffs, size_t n_coeffs, size_t log_n, const uint64_t *gen,
                         const uint64_t *shift, uint64_t *d_out)
{
    return iopx_mul_fft_fp3_windows_dev(d_coeffs, n_coeffs, log_n, gen, shift, d_out, 0, nullptr, nullptr, nullptr);
}

int iopx_mul_fft_fp3_windows_dev(const uint64_t *d_coeffs, size_t n_coeffs, size_t log_n, const uint64_t *gen, const uint64_t *shift, uint64_t *d_out,
                                 size_t num_windows, const size_t *window_first, const size_t *window_log_stride, uint64_t *const *d_windows)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    if (!gen || !shift || !d_out || (n_coeffs && !d_coeffs)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const size_t n = (size_t)1 << log_n;
    if (n_coeffs > n) return fail(IOPX_ERR_INVALID_ARGUMENT, "multiplicative FFT: %zu coefficients exceed the domain size %zu", n_coeffs, n);
    if (num_windows > 2) return fail(IOPX_ERR_INVALID_ARGUMENT, "at most two windows per transform");
    MfWindows wins;
    for (size_t w = 0; w < num_windows; ++w) {
        if (!window_first || !window_log_stride || !d_windows || !d_windows[w]) return fail(IOPX_ERR_INVALID_ARGUMENT, "null window argument");
        if (window_log_stride[w] > log_n || window_first[w] >= ((size_t)1 << window_log_stride[w])) return fail(IOPX_ERR_INVALID_ARGUMENT, "window %zu: first %zu, stride 2^%zu of a 2^%zu-point domain", w, window_first[w], window_log_stride[w], log_n);
        if (d_windows[w] == d_out) return fail(IOPX_ERR_INVALID_ARGUMENT, "a window may not alias the output");
        wins.dst[w] = d_windows[w]; wins.first[w] = (uint32_t)window_first[w]; wins.log_stride[w] = (int)window_log_stride[w];
    }
    wins.num = (int)num_windows;
    if (n_coeffs == 0) {
        { const int crc_ = iopx::fill_bytes(d_out, 0, n * 24); if (crc_ != IOPX_OK) return crc_; }
        for (size_t w = 0; w < num_windows; ++w) { const int crc_ = iopx::fill_bytes(d_windows[w], 0, (n >> window_log_stride[w]) * 24); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }
    MulPlan *pl = nullptr;
    rc = get_mplan((int)log_n, gen, &pl);
    if (rc != IOPX_OK) return rc;
    if (!pl->have_fwd && (rc = build_cache(*pl, false)) != IOPX_OK) return rc;
    const int logd = (int)ceil_log2(n_coeffs);
    const hfp3 sh = hfp3::from_words(shift);
    if (sh.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "multiplicative FFT: zero coset shift");
    TmpBuf scaled, hi, lo;
    const uint64_t *src = d_coeffs;
    if (d_coeffs == d_out) {    // the first pass permutes: it cannot run in place
        if ((rc = scaled.alloc(n_coeffs * 24)) != IOPX_OK) return rc;
        { const int crc_ = iopx::copy_d2d(scaled.p, d_coeffs, n_coeffs * 24); if (crc_ != IOPX_OK) return crc_; }
        src = scaled.u64();
    }
    if (!(sh == hfp3::one()) && n_coeffs > 1) {
        if (!scaled.p && (rc = scaled.alloc(n_coeffs * 24)) != IOPX_OK) return rc;
        if ((rc = build_two_level(sh, hfp3::one(), logd, hi, lo)) != IOPX_OK) return rc;
        { ProfScope ps_("k_fp_scale_pow"); hipLaunchKernelGGL(k_fp_scale_pow, dim3(mgrid(n_coeffs, 256)), dim3(256), 0, stream(), scaled.u64(), src, (const uint64_t *)hi.u64(), (const uint64_t *)lo.u64(), n_coeffs); }
        src = scaled.u64();
    }
    rc = run_mfft(pl->cache_fwd.u64(), src, n_coeffs, d_out, (int)log_n, (int)log_n - logd, 0, nullptr, nullptr, &wins);
    if (rc != IOPX_OK) return rc;
    return IOPX_OK;                                     // per-call tables are released in stream order
}

// ---- host-side scalars of the prime field (domain metadata of the template boundary; no device needed) ----
// multiplicative_subgroup_base::construct_internal (subgroup.tcc:55-59): multiplicative_generator^((p - 1) / 2^log_order)
int iopx_fp3_subgroup_generator(size_t log_order, uint64_t *gen)
{
    if (!gen) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (log_order > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_order %zu exceeds the 2-adicity of the field", log_order);
    uint64_t e[3] = { hfp3::P[0] - 1, hfp3::P[1], hfp3::P[2] };
    for (size_t s = 0; s < log_order; ++s) { e[0] = (e[0] >> 1) | (e[1] << 63); e[1] = (e[1] >> 1) | (e[2] << 63); e[2] >>= 1; }
    const hfp3 g = hfp3::from_uint(19).pow_limbs(e, 3);
    memcpy(gen, g.w, 24);
    return IOPX_OK;
}
int iopx_fp3_multiplicative_generator(uint64_t *gen)           // libff edwards_Fr::multiplicative_generator = 19 (recalled)
{
    if (!gen) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hfp3 g = hfp3::from_uint(19);
    memcpy(gen, g.w, 24);
    return IOPX_OK;
}
int iopx_fp3_host_mul(const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    if (!a || !b || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hfp3 r = hfp3::from_words(a) * hfp3::from_words(b);
    memcpy(out, r.w, 24);
    return IOPX_OK;
}
int iopx_fp3_host_pow(const uint64_t *a, uint64_t exponent, uint64_t *out)
{
    if (!a || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hfp3 r = hfp3::from_words(a).pow(exponent);
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

// d_out[l] = init * base^l for l < count, as ordinary libff elements (multi_lincheck's alpha powers, basic_lincheck_aux.tcc:37-45)
int iopx_fp3_pow_table_dev(uint64_t *d_out, size_t count, const uint64_t *base, const uint64_t *init)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!d_out || !base || !init) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (count == 0) return IOPX_OK;
    const int nb = (int)ceil_log2(count);
    // fp_build_pow writes multipliers (x 2^203); data are x 2^192: start from init 2^-11
    const hfp3 init_d = hfp3::from_words(init) * hfp3::from_uint(2048).inverse();
    if (((size_t)1 << nb) == count) return fp_build_pow(d_out, hfp3::from_words(base), init_d, nb);
    TmpBuf full;
    if ((rc = full.alloc((((size_t)1) << nb) * 24)) != IOPX_OK) return rc;
    if ((rc = fp_build_pow(full.u64(), hfp3::from_words(base), init_d, nb)) != IOPX_OK) return rc;
    { const int crc_ = iopx::copy_d2d(d_out, full.p, count * 24); if (crc_ != IOPX_OK) return crc_; }
    return IOPX_OK;
}

int iopx_mul_ifft_fp3_dev(const uint64_t *d_evals, size_t log_n, const uint64_t *gen, const uint64_t *shift, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    if (!gen || !shift || !d_out || !d_evals) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const size_t n = (size_t)1 << log_n;
    if (log_n == 0) {       // multiplicative_IFFT_wrapper returns {v[0]} for size 1 (fft.tcc:397-401)
        { const int crc_ = iopx::copy_d2d(d_out, d_evals, 24); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }
    MulPlan *pl = nullptr;
    rc = get_mplan((int)log_n, gen, &pl);
    if (rc != IOPX_OK) return rc;
    if (!pl->have_inv && (rc = build_cache(*pl, true)) != IOPX_OK) return rc;
    const hfp3 sh = hfp3::from_words(shift);
    if (sh.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "multiplicative IFFT: zero coset shift");
    const hfp3 ninv = hfp3::from_uint((uint64_t)n).inverse();
    TmpBuf hi, lo, tmp;
    int scale = 1;
    if (sh == hfp3::one()) {
        if ((rc = hi.alloc(24)) != IOPX_OK) return rc;
        const hfp3 ninv_t = ninv.table_form();
        { int urc_ = upload(hi.p, ninv_t.w, 24); if (urc_ != IOPX_OK) return urc_; }
    } else {
        scale = 2;
        if ((rc = build_two_level(sh.inverse(), ninv, (int)log_n, hi, lo)) != IOPX_OK) return rc;
    }
    const uint64_t *src = d_evals;
    if (d_evals == d_out) {     // the first pass permutes: it cannot run in place
        if ((rc = tmp.alloc(n * 24)) != IOPX_OK) return rc;
        { const int crc_ = iopx::copy_d2d(tmp.p, d_evals, n * 24); if (crc_ != IOPX_OK) return crc_; }
        src = tmp.u64();
    }
    rc = run_mfft(pl->cache_inv.u64(), src, n, d_out, (int)log_n, 0, scale, hi.u64(), lo.p ? lo.u64() : nullptr);
    if (rc != IOPX_OK) return rc;
    return IOPX_OK;
}

int iopx_mul_ifft_known_degree_fp3_dev(const uint64_t *d_evals, size_t degree, size_t log_n, const uint64_t *gen,
                                       const uint64_t *shift, uint64_t *d_out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31 || degree == 0 || degree > ((size_t)1 << log_n)) return fail(IOPX_ERR_INVALID_ARGUMENT, "bad degree / domain size");
    const int k = (int)ceil_log2(degree);
    const size_t pow2 = (size_t)1 << k, stride = ((size_t)1 << log_n) >> k;
    TmpBuf sub;
    if ((rc = sub.alloc(pow2 * 24)) != IOPX_OK) return rc;
    { ProfScope ps_("k_fp_gather_stride"); hipLaunchKernelGGL(k_fp_gather_stride, dim3(mgrid(3 * pow2, 256)), dim3(256), 0, stream(), sub.u64(), d_evals, stride, pow2); }
    // generator of the sub-coset: g^(n / pow2)
    hfp3 gs = hfp3::from_words(gen);
    for (size_t s = stride; s > 1; s >>= 1) gs = gs.squared();
    rc = iopx_mul_ifft_fp3_dev(sub.u64(), (size_t)k, gs.w, shift, d_out);
    if (rc != IOPX_OK) return rc;
    return IOPX_OK;
}

int iopx_fri_fold_mul_fp3_dev(const uint64_t *d_f_i, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                              size_t coset_size, const uint64_t *x_i, uint64_t *d_next)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    if (!d_f_i || !d_next || !gen || !shift || !x_i) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (coset_size == 0 || (coset_size & (coset_size - 1))) return fail(IOPX_ERR_INVALID_ARGUMENT, "coset size %zu is not a power of two", coset_size);
    const int eta = (int)ceil_log2(coset_size);
    if ((size_t)eta > log_n) return fail(IOPX_ERR_INVALID_ARGUMENT, "coset size %zu exceeds the domain size", coset_size);
    const size_t n = (size_t)1 << log_n;
    if (eta == 0) {
        { const int crc_ = iopx::copy_d2d(d_next, d_f_i, n * 24); if (crc_ != IOPX_OK) return crc_; }
        return IOPX_OK;
    }
    MulPlan *pl = nullptr;
    rc = get_mplan((int)log_n, gen, &pl);
    if (rc != IOPX_OK) return rc;
    if (!pl->have_inv && (rc = build_cache(*pl, true)) != IOPX_OK) return rc;
    hfp3 sh = hfp3::from_words(shift), x = hfp3::from_words(x_i);
    if (sh.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "FRI fold: zero coset shift");
    const hfp3 inv2 = hfp3::from_uint(2).inverse().table_form();
    std::vector<uint64_t> hc;
    for (int e = 0; e < eta; ++e) {
        const hfp3 xs = (x * sh.inverse()).table_form();
        hc.insert(hc.end(), xs.w, xs.w + 3);
        hc.insert(hc.end(), inv2.w, inv2.w + 3);
        sh = sh.squared();
        x = x.squared();
    }
    TmpBuf dc;
    if ((rc = dc.alloc(hc.size() * 8)) != IOPX_OK) return rc;
    { int urc_ = upload(dc.p, hc.data(), hc.size() * 8); if (urc_ != IOPX_OK) return urc_; }
    const uint64_t *ginv_top = pl->cache_inv.u64() + 3 * ((n >> 1) - 1);
    if (eta <= 3) {
        MfoldParams p;
        p.src = d_f_i; p.dst = d_next; p.ginv = ginv_top; p.consts = dc.u64(); p.half = n >> eta; p.stride_log = 0;
        const size_t bytes = (n + p.half) * 24;
        if (eta == 1) { ProfScope ps_("k_fri_fold_fused_mul_eta1", bytes); hipLaunchKernelGGL(k_fri_fold_fused_mul<1>, dim3(mgrid(p.half, 256)), dim3(256), 0, stream(), p); }
        else if (eta == 2) { ProfScope ps_("k_fri_fold_fused_mul_eta2", bytes); hipLaunchKernelGGL(k_fri_fold_fused_mul<2>, dim3(mgrid(p.half, 256)), dim3(256), 0, stream(), p); }
        else { ProfScope ps_("k_fri_fold_fused_mul_eta3", bytes); hipLaunchKernelGGL(k_fri_fold_fused_mul<3>, dim3(mgrid(p.half, 256)), dim3(256), 0, stream(), p); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    }
    TmpBuf tmp[2];
    const uint64_t *src = d_f_i;
    size_t cur = n;
    for (int e = 0; e < eta; ++e) {
        const size_t half = cur >> 1;
        uint64_t *dst = d_next;
        if (e != eta - 1) {
            if ((rc = tmp[e & 1].alloc(half * 24)) != IOPX_OK) return rc;
            dst = tmp[e & 1].u64();
        }
        MfoldParams p;
        p.src = src; p.dst = dst; p.ginv = ginv_top; p.consts = dc.u64() + 6 * e; p.half = half; p.stride_log = e;
        { ProfScope ps_("k_fri_fold2_mul"); hipLaunchKernelGGL(k_fri_fold2_mul, dim3(mgrid(half, 256)), dim3(256), 0, stream(), p); }
        src = dst;
        cur = half;
    }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// ---- host-pointer variants ---------------------------------------------------------------------
int iopx_mul_fft_fp3(const uint64_t *coeffs, size_t n_coeffs, size_t log_n, const uint64_t *gen, const uint64_t *shift, uint64_t *out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    const size_t n = (size_t)1 << log_n;
    if (n_coeffs > n) return fail(IOPX_ERR_INVALID_ARGUMENT, "multiplicative FFT: %zu coefficients exceed the domain size %zu", n_coeffs, n);
    DevBuf din, dout;
    if ((rc = din.alloc(n_coeffs * 24)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(n * 24)) != IOPX_OK) return rc;
    if (n_coeffs) IOPX_HIP(copy_h2d(din.p, coeffs, n_coeffs * 24, stream()));
    if ((rc = iopx_mul_fft_fp3_dev(din.u64(), n_coeffs, log_n, gen, shift, dout.u64())) != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(out, dout.p, n * 24, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

int iopx_mul_ifft_fp3(const uint64_t *evals, size_t log_n, const uint64_t *gen, const uint64_t *shift, uint64_t *out)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    const size_t n = (size_t)1 << log_n;
    DevBuf din, dout;
    if ((rc = din.alloc(n * 24)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(n * 24)) != IOPX_OK) return rc;
    IOPX_HIP(copy_h2d(din.p, evals, n * 24, stream()));
    if ((rc = iopx_mul_ifft_fp3_dev(din.u64(), log_n, gen, shift, dout.u64())) != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(out, dout.p, n * 24, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

int iopx_fri_fold_mul_fp3(const uint64_t *f_i, size_t log_n, const uint64_t *gen, const uint64_t *shift, size_t coset_size,
                          const uint64_t *x_i, uint64_t *next)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (log_n > 31) return fail(IOPX_ERR_INVALID_ARGUMENT, "log_n %zu exceeds the 2-adicity of the field", log_n);
    if (coset_size == 0 || (coset_size & (coset_size - 1)) || coset_size > ((size_t)1 << log_n))
        return fail(IOPX_ERR_INVALID_ARGUMENT, "bad coset size %zu", coset_size);
    const size_t n = (size_t)1 << log_n, n_out = n / coset_size;
    DevBuf din, dout;
    if ((rc = din.alloc(n * 24)) != IOPX_OK) return rc;
    if ((rc = dout.alloc(n_out * 24)) != IOPX_OK) return rc;
    IOPX_HIP(copy_h2d(din.p, f_i, n * 24, stream()));
    if ((rc = iopx_fri_fold_mul_fp3_dev(din.u64(), log_n, gen, shift, coset_size, x_i, dout.u64())) != IOPX_OK) return rc;
    IOPX_HIP(copy_d2h(next, dout.p, n_out * 24, stream()));
    IOPX_HIP(hipStreamSynchronize(stream()));
    return IOPX_OK;
}

} // extern "C"
