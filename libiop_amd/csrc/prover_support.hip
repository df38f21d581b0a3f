// What a native (C++) prover built on the C ABI needs next to the kernels — round 3, row (b') of the scope table:
//   * pooled, stream-ordered device allocations for oracles and trees (the C++ mirror's device_vector; a Python caller has
//     torch's caching allocator for this),
//   * element gather / scatter by an index vector and device-to-device copies (field_subset::reindex_by_subset applied to whole
//     vectors: libiop/protocols/encoded/lincheck/basic_lincheck_aux.tcc:50-58, r1cs_rs_iop.tcc:406-430),
//   * host BLAKE2b (RFC 7693) for the Fiat-Shamir hashchain of libiop/bcs/hashing/blake2b.tcc:10-110 — 32-byte states and
//     O(log n) challenges, never codeword-sized data,
//   * host scalar arithmetic in both fields for the handful of per-proof constants,
//   * byte counters of every host<->device copy, so a caller can assert that no codeword crossed PCIe.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include "runtime.h"
#include "gf192_host.h"
#include "fp3_host.h"

namespace iopx {

// ---- pooled allocations ---------------------------------------------------------------------------------------------------------
static std::map<void *, size_t> g_pool_caps;        // live pooled blocks -> their capacity (what tmp_free needs back)
static std::mutex g_pool_mu;

// ---- transfer accounting (runtime.hip adds to these) -------------------------------------------------------------------------
std::atomic_uint_fast64_t g_bytes_h2d{0}, g_bytes_d2h{0};

// ---- gather / scatter -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_words(const uint64_t *src, const uint64_t *index, size_t count, int words, uint64_t *dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t *s = src + (size_t)words * index[i];
        for (int w = 0; w < words; ++w) dst[(size_t)words * i + w] = s[w];
    }
}

// dst[i] = src[i * stride]: one lane per 8-byte word, so the writes are contiguous
__global__ void __launch_bounds__(256) k_gather_stride_words(const uint64_t *src, size_t count, size_t stride, int words, uint64_t *dst)
{
    const size_t total = count * (size_t)words;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / (size_t)words, w = t % (size_t)words;
        dst[t] = src[i * stride * (size_t)words + w];
    }
}

// *count += the number of 8-byte words at which a and b differ (an atomic only from lanes that saw a difference: none, in the expected case)
__global__ void __launch_bounds__(256) k_count_mismatch_words(const uint64_t *a, const uint64_t *b, size_t words, unsigned long long *count)
{
    unsigned long long mine = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < words; t += (size_t)gridDim.x * blockDim.x) mine += a[t] != b[t];
    if (mine) atomicAdd(count, mine);
}

__global__ void __launch_bounds__(256) k_scatter_words(const uint64_t *src, const uint64_t *index, size_t count, int words, uint64_t *dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t *d = dst + (size_t)words * index[i];
        for (int w = 0; w < words; ++w) d[w] = src[(size_t)words * i + w];
    }
}

static unsigned grid_of(size_t count)
{
    size_t g = (count + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// ---- BLAKE2b on the host (RFC 7693) --------------------------------------------------------------------------------------------
static const uint64_t HB2B_IV[8] = {
    0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
    0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };
static const uint8_t HB2B_SIGMA[12][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 } };

static inline uint64_t hrotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void hb2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last)
{
    uint64_t m[16], v[16];
    memcpy(m, block, 128);                                  // little-endian host
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = HB2B_IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = HB2B_SIGMA[r];
#define HG(a, b, c, d, x, y)                                                   \
        v[a] = v[a] + v[b] + (x); v[d] = hrotr(v[d] ^ v[a], 32);               \
        v[c] = v[c] + v[d];       v[b] = hrotr(v[b] ^ v[c], 24);               \
        v[a] = v[a] + v[b] + (y); v[d] = hrotr(v[d] ^ v[a], 16);               \
        v[c] = v[c] + v[d];       v[b] = hrotr(v[b] ^ v[c], 63);
        HG(0, 4, 8, 12, m[s[0]], m[s[1]]) HG(1, 5, 9, 13, m[s[2]], m[s[3]]) HG(2, 6, 10, 14, m[s[4]], m[s[5]]) HG(3, 7, 11, 15, m[s[6]], m[s[7]])
        HG(0, 5, 10, 15, m[s[8]], m[s[9]]) HG(1, 6, 11, 12, m[s[10]], m[s[11]]) HG(2, 7, 8, 13, m[s[12]], m[s[13]]) HG(3, 4, 9, 14, m[s[14]], m[s[15]])
#undef HG
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

static int host_blake2b(uint8_t *out, size_t outlen, const uint8_t *msg, size_t msglen, const uint8_t *key, size_t keylen)
{
    if (outlen == 0 || outlen > 64 || keylen > 64) return fail(IOPX_ERR_INVALID_ARGUMENT, "blake2b: digest length 1..64, key length <= 64");
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = HB2B_IV[i];
    h[0] ^= 0x01010000ull ^ ((uint64_t)keylen << 8) ^ (uint64_t)outlen;
    uint8_t block[128];
    uint64_t t = 0;
    size_t remaining = msglen;
    if (keylen) {                                           // the key is the first block, zero padded
        memset(block, 0, 128);
        memcpy(block, key, keylen);
        t = 128;
        if (msglen == 0) { hb2b_compress(h, block, t, true); memcpy(out, h, outlen); return IOPX_OK; }
        hb2b_compress(h, block, t, false);
    }
    while (remaining > 128) {
        t += 128;
        hb2b_compress(h, msg, t, false);
        msg += 128; remaining -= 128;
    }
    memset(block, 0, 128);
    if (remaining) memcpy(block, msg, remaining);
    t += remaining;
    hb2b_compress(h, block, t, true);
    memcpy(out, h, outlen);
    return IOPX_OK;
}

} // namespace iopx

using namespace iopx;

extern "C" {

int iopx_pool_alloc(void **dptr, size_t bytes)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!dptr) return fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_pool_alloc: null out pointer");
    size_t cap = 0;
    void *p = tmp_alloc(bytes ? bytes : 8, &cap);
    if (!p) return fail(IOPX_ERR_RUNTIME, "device allocation of %zu bytes failed", bytes);
    { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool_caps[p] = cap; }
    *dptr = p;
    return IOPX_OK;
}

int iopx_pool_free(void *dptr)
{
    if (!dptr) return IOPX_OK;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_caps.find(dptr);
        if (it == g_pool_caps.end()) return fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_pool_free: not a pooled block");
        cap = it->second;
        g_pool_caps.erase(it);
    }
    tmp_free(dptr, cap);                    // reuse is ordered by the library's stream (runtime.h)
    return IOPX_OK;
}

int iopx_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (bytes) { const int crc_ = iopx::copy_d2d(dst_dev, src_dev, bytes); if (crc_ != IOPX_OK) return crc_; }
    return IOPX_OK;
}

int iopx_memset_dev(void *dst_dev, int value, size_t bytes)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (bytes) { const int crc_ = iopx::fill_bytes(dst_dev, value, bytes); if (crc_ != IOPX_OK) return crc_; }
    return IOPX_OK;
}

int iopx_upload_small(void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    return upload(dst_dev, src_host, bytes);
}

int iopx_gather_dev(const void *d_src, const uint64_t *d_index, size_t count, size_t elem_bytes, void *d_dst)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (elem_bytes == 0 || elem_bytes % 8) return fail(IOPX_ERR_INVALID_ARGUMENT, "gather: element size must be a multiple of 8 bytes");
    if (count == 0) return IOPX_OK;
    { ProfScope ps_("k_gather_words"); hipLaunchKernelGGL(k_gather_words, dim3(grid_of(count)), dim3(256), 0, stream(), (const uint64_t *)d_src, d_index, count, (int)(elem_bytes / 8), (uint64_t *)d_dst); }
    return IOPX_OK;
}

int iopx_gather_stride_dev(const void *d_src, size_t count, size_t stride, size_t elem_bytes, void *d_dst)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (elem_bytes == 0 || elem_bytes % 8) return fail(IOPX_ERR_INVALID_ARGUMENT, "gather: element size must be a multiple of 8 bytes");
    if (stride == 0) return fail(IOPX_ERR_INVALID_ARGUMENT, "gather: stride must be positive");
    if (count == 0) return IOPX_OK;
    if (!d_src || !d_dst) return fail(IOPX_ERR_INVALID_ARGUMENT, "gather: null argument");
    const size_t words = elem_bytes / 8;
    { ProfScope ps_("k_gather_stride_words", 2 * count * elem_bytes); hipLaunchKernelGGL(k_gather_stride_words, dim3(grid_of(count * words)), dim3(256), 0, stream(), (const uint64_t *)d_src, count, stride, (int)words, (uint64_t *)d_dst); }
    return IOPX_OK;
}

int iopx_count_mismatch_dev(const void *d_a, const void *d_b, size_t bytes, uint64_t *d_count)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (bytes % 8) return fail(IOPX_ERR_INVALID_ARGUMENT, "count_mismatch: the byte count must be a multiple of 8");
    if (bytes == 0) return IOPX_OK;
    if (!d_a || !d_b || !d_count) return fail(IOPX_ERR_INVALID_ARGUMENT, "count_mismatch: null argument");
    { ProfScope ps_("k_count_mismatch_words", 2 * bytes); hipLaunchKernelGGL(k_count_mismatch_words, dim3(grid_of(bytes / 8)), dim3(256), 0, stream(), (const uint64_t *)d_a, (const uint64_t *)d_b, bytes / 8, (unsigned long long *)d_count); }
    return IOPX_OK;
}

int iopx_scatter_dev(const void *d_src, const uint64_t *d_index, size_t count, size_t elem_bytes, void *d_dst)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (elem_bytes == 0 || elem_bytes % 8) return fail(IOPX_ERR_INVALID_ARGUMENT, "scatter: element size must be a multiple of 8 bytes");
    if (count == 0) return IOPX_OK;
    { ProfScope ps_("k_scatter_words"); hipLaunchKernelGGL(k_scatter_words, dim3(grid_of(count)), dim3(256), 0, stream(), (const uint64_t *)d_src, d_index, count, (int)(elem_bytes / 8), (uint64_t *)d_dst); }
    return IOPX_OK;
}

int iopx_transfer_stats(uint64_t *h2d_bytes, uint64_t *d2h_bytes, int reset)
{
    if (h2d_bytes) *h2d_bytes = g_bytes_h2d.load();
    if (d2h_bytes) *d2h_bytes = g_bytes_d2h.load();
    if (reset) { g_bytes_h2d = 0; g_bytes_d2h = 0; }
    return IOPX_OK;
}

int iopx_blake2b_host(uint8_t *out, size_t outlen, const void *msg, size_t msglen, const void *key, size_t keylen)
{
    if (!out || (!msg && msglen) || (!key && keylen)) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    return host_blake2b(out, outlen, (const uint8_t *)msg, msglen, (const uint8_t *)key, keylen);
}

int iopx_gf192_host_mul(const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    if (!a || !b || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hgf192 r = hgf192::from_words(a) * hgf192::from_words(b);
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

int iopx_fp3_host_add(const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    if (!a || !b || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    hfp3 nb;                                                 // a + b = a - (0 - b)
    const hfp3 r = hfp3::from_words(a) - (nb - hfp3::from_words(b));
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

int iopx_fp3_host_sub(const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    if (!a || !b || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hfp3 r = hfp3::from_words(a) - hfp3::from_words(b);
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

int iopx_fp3_host_inverse(const uint64_t *a, uint64_t *out)
{
    if (!a || !out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hfp3 r = hfp3::from_words(a).inverse();
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

int iopx_fp3_from_uint(uint64_t value, uint64_t *out)
{
    if (!out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    const hfp3 r = hfp3::from_uint(value);
    memcpy(out, r.w, 24);
    return IOPX_OK;
}

int iopx_fp3_modulus(uint64_t *out)
{
    if (!out) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, hfp3::P, 24);
    return IOPX_OK;
}

} // extern "C"
