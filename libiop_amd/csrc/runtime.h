// Internal runtime of libiop_amd: error reporting, the current HIP stream, small helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include <memory>
#include <string>
#include "../../include/libiop_amd.h"

namespace iopx {

void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

// Current stream for all enqueued work (iopx_set_stream); never null after ensure_device().
hipStream_t stream();

// Returns IOPX_OK when a HIP device is usable (and binds the library's stream on first use).
int ensure_device();

#define IOPX_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::iopx::fail(IOPX_ERR_RUNTIME, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Every host<->device copy of the library goes through these two, which keep byte counters (iopx_transfer_stats): a prover that
// claims "no codeword crosses PCIe" can be checked against them.
extern std::atomic_uint_fast64_t g_bytes_h2d, g_bytes_d2h;
static inline hipError_t copy_h2d(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    g_bytes_h2d += bytes;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
}
static inline hipError_t copy_d2h(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    g_bytes_d2h += bytes;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
}

// Device buffer owned by a plan / a call (freed in the destructor, stream-ordered use only).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n)
    {
        release();
        if (n == 0) n = 8;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return fail(IOPX_ERR_RUNTIME, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
        bytes = n;
        return IOPX_OK;
    }
    void release()
    {
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    }
    uint64_t *u64() const { return (uint64_t *)p; }
};

// Stream-ordered upload of a small host array (per-call constants): the bytes are copied into a pinned staging
// chunk owned by the library, so the caller's buffer may be freed on return and the copy needs no synchronisation.
int upload(void *dst_dev, const void *src_host, size_t bytes);
// Blocking read-back of a small result (roots, proof-of-work answers, query responses, membership paths) through a pinned bounce buffer:
// queued behind the stream's work, then the stream is drained.  A pageable destination would make the runtime stage the copy itself, at
// several times the latency — and a proof has some thirty of these on its critical path.
int download(void *dst_host, const void *src_dev, size_t bytes, bool deferrable = false);
int defer_downloads_begin();
int defer_downloads_end();
// Between iopx_defer_downloads_begin() and _end(), deferrable read-backs (transcript extraction: query responses, membership paths) are only
// queued; _end() drains the stream once and delivers all of them — one synchronisation for the whole query phase instead of two per tree.

// Per-call temporary in device memory.  All work of the library is enqueued on ONE stream in program order, so a
// temporary released by one call may be handed to a later call without any synchronisation: the later call's
// kernels run after the earlier call's kernels on that stream.  Blocks are cached in a free list (runtime.hip)
// and returned to HIP by iopx_clear_plans(); iopx_set_stream() synchronises before switching streams.
void *tmp_alloc(size_t bytes, size_t *cap);
void tmp_free(void *p, size_t cap);
void tmp_trim();

struct TmpBuf {
    void *p = nullptr;
    size_t bytes = 0, cap = 0;
    TmpBuf() {}
    TmpBuf(const TmpBuf &) = delete;
    TmpBuf &operator=(const TmpBuf &) = delete;
    ~TmpBuf() { release(); }
    int alloc(size_t n)
    {
        release();
        if (n == 0) n = 8;
        p = tmp_alloc(n, &cap);
        if (!p) return fail(IOPX_ERR_RUNTIME, "device allocation of %zu bytes failed", n);
        bytes = n;
        return IOPX_OK;
    }
    void release()
    {
        if (!p) return;
        if (!borrowed) tmp_free(p, cap);
        p = nullptr; bytes = 0; cap = 0; borrowed = false;
        keep.reset();
    }
    // points at device memory owned elsewhere (a cached table): released without being returned to the pool.  `owner` keeps the
    // table alive for as long as this call holds it, so a cache eviction between the borrow and the kernel launch cannot free it
    // (hipFree in the last owner's destructor waits for the device, so kernels in flight are safe as well)
    void borrow(void *q, size_t n, std::shared_ptr<DevBuf> owner = nullptr)
    {
        release();
        p = q; bytes = n; cap = 0; borrowed = true;
        keep = std::move(owner);
    }
    bool borrowed = false;
    std::shared_ptr<DevBuf> keep;
    uint64_t *u64() const { return (uint64_t *)p; }
};

// Device-resident copies of host-built tables that depend on DOMAINS only (subset-sum tables of vanishing-polynomial values, inverse tables per coset,
// power tables): a proof needs the same ones as the proof before it, and building + uploading them (25 KB per table) sat on the critical path
// between rounds.  `build` fills the host words on a miss; the entry stays alive while a TmpBuf borrows it (TmpBuf::borrow's owner).
int cached_domain_table(const std::vector<uint64_t> &key, const std::function<int(std::vector<uint64_t> &)> &build, TmpBuf &out);
void clear_domain_tables();

// Per-kernel timing with HIP events on the library's stream (iopx_profile_begin / iopx_profile_report).
// Costs nothing when profiling is off.  Usage: { ProfScope ps("k_name"); hipLaunchKernelGGL(...); }
// work_bytes: the launch's ALGORITHMIC bytes (elements swept x 24 x (read + write)), summed per kernel in the report;
// work_products: the field multiplications the launch performs (the unit of the ALU ceiling in bench.py's roofline block).
struct ProfScope {
    int slot;
    explicit ProfScope(const char *name, size_t work_bytes = 0, size_t work_products = 0);
    ~ProfScope();
};

// fft_mul.hip: two-level power tables of the 181-bit prime field on the device,
// hi[q] = init * base^(4096 q) (q < 2^max(logc-12,0)), lo[r] = base^r (r < 4096), so init * base^j = hi[j >> 12] * lo[j & 4095]
struct hfp3;
// Tables are kept by (base, init, logc) — most of them depend on the domain only and recur in every proof; cache_hi = false for a
// table whose init carries a per-proof value (the LDT coefficients): its lo half is still shared by base.
int build_two_level(const hfp3 &base, const hfp3 &init, int logc, TmpBuf &hi, TmpBuf &lo, bool cache_hi = true);

// Per-module plan caches dropped by iopx_clear_plans() (the caller has synchronised the device).
// Options: named integers that select a schedule or a tile geometry.  One table for the whole library: a name's value is whatever
// iopx_set_option gave it, otherwise the environment variable of that name read ONCE (at the name's first lookup: the library's only getenv),
// otherwise the caller's default.  Tile geometries are latched by their component at its first use; schedule options are looked up per proof
// (a table lookup), so a test or a bench can switch them between two proofs with iopx_set_option.
int opt(const char *name, int dflt);
int opt_range(const char *name, int dflt, int lo, int hi);      // clamped

// Device-to-device copies and fills as KERNELS on the active stream (hipMemcpyAsync / hipMemsetAsync leave the GPU idle for several microseconds
// on both sides of the runtime's own copy kernel: profiles/r05_gpu_gaps.txt).  Any alignment, any size.
int copy_d2d(void *dst_dev, const void *src_dev, size_t bytes);
int fill_bytes(void *dst_dev, int value, size_t bytes);

// One-time ("cold") host-side costs — device allocations, plan and table builds, matrix transpositions — accumulated per label as wall time
// (iopx_cold_stats): what a process's first proof pays beyond its kernels, and what iopx_aurora_instance_warm moves out of it.
struct ColdScope {
    const char *label;
    long long t0_ns;
    explicit ColdScope(const char *label);
    ~ColdScope();
};
void cold_add(const char *label, double ms);

// the side stream (runtime.hip): the provers' Merkle trees (C ABI iopx_side_stream_*)
int side_stream_fork(int k);
int side_stream_select(int k);      // -1: back to the main stream
int side_stream_join(int k);
int side_stream_current();
void clear_mul_plans();
void clear_dist_plans();           // fft_add_dist.hip: the sharded transforms' per-rank twist tables
void clear_poseidon_sets();

// comm.hip: the communicator bound for transforms (iopx_comm_bind_transforms), or null; its rank / world; collectives for library-internal
// use (same semantics as the C entry points)
struct CommInfo { iopx_comm *comm; int rank, world; };
CommInfo transform_comm();

static inline size_t ceil_log2(size_t n)
{
    size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) { n >>= 1; ++r; }
    return r;
}

} // namespace iopx
