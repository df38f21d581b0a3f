// libiop_amd runtime: device binding, stream, memory helpers, error strings (see include/libiop_amd.h).
#include "runtime.h"
#include <algorithm>
#include <atomic>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace iopx {

static thread_local char g_err[512] = "";
static hipStream_t g_stream = nullptr;      // stream in use (may legitimately be 0: the HIP legacy default stream)
static hipStream_t g_own_stream = nullptr;  // created lazily
static bool g_caller_stream = false;        // g_stream was chosen by iopx_set_stream
static bool g_ready = false;
static int g_device = -1;                   // the device this process is bound to (one process per GPU)
static std::mutex g_mu;

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

// ---- the side stream (round 5): work that the next launches on the main stream do not wait for runs beside them — a round's Merkle tree,
// whose root only the transcript needs (bcs_prover::finish_round).  (A second one, for every other coset group of a low-degree extension, was
// tried: independent chains of launches, each filling the other's launch tails — 1893 -> 1704 cycles per wave-butterfly in tools/ubench/bfly_loop,
// nothing in the prover, 37.98 -> 38.3 ms: profiles/r05_ab_fft_streams.txt.)
// fork: the side stream waits for everything the main stream holds at that moment; select: which stream the library's launches, copies and
// pool frees go to; join: the main stream waits for the side stream (no host wait) and the blocks freed on it become reusable.  One host
// thread per process drives them, like the rest of the prover entry points. ----
#define IOPX_SIDE_STREAMS 1
static hipStream_t g_side_stream[IOPX_SIDE_STREAMS] = { nullptr };
static hipEvent_t g_fork_event[IOPX_SIDE_STREAMS], g_join_event[IOPX_SIDE_STREAMS];
static std::atomic<int> g_cur_side{ -1 };   // -1: the main stream is current (written by the driving thread under g_mu; read by pool frees of any thread)
static bool g_side_elided = false;          // a begin/end pair that stayed on the main stream because the HIP-event profiler records (kernels timed alone)
static bool g_side_dirty[IOPX_SIDE_STREAMS] = { false };      // the side stream holds work the main stream has not waited for
static inline hipStream_t active_stream() { return g_cur_side >= 0 ? g_side_stream[g_cur_side] : g_stream; }

// cached device temporaries (single-stream reuse, see runtime.h): the free list, and what was freed while a side stream was current — reusable
// once the main stream has joined it
struct TmpBlock { void *p; size_t cap; };
static std::vector<TmpBlock> g_tmp_free;
static std::mutex g_tmp_mu;
static std::vector<TmpBlock> g_tmp_quarantine[IOPX_SIDE_STREAMS];

hipStream_t stream() { std::lock_guard<std::mutex> lk(g_mu); return active_stream(); }
int bound_device();

int ensure_device()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready) {
        // a host thread other than the binding one starts on device 0: every entry point runs on the bound device
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != g_device) {
            const hipError_t e = hipSetDevice(g_device);
            if (e != hipSuccess) return fail(IOPX_ERR_RUNTIME, "hipSetDevice(%d) failed: %s", g_device, hipGetErrorString(e));
        }
        return IOPX_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(IOPX_ERR_NO_DEVICE, "no HIP device available (%s); libiop_amd has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (!g_own_stream) {
        e = hipStreamCreateWithFlags(&g_own_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return fail(IOPX_ERR_RUNTIME, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    if (!g_caller_stream) g_stream = g_own_stream;
    if (hipGetDevice(&g_device) != hipSuccess) return fail(IOPX_ERR_RUNTIME, "hipGetDevice failed");
    g_ready = true;
    return IOPX_OK;
}

int bound_device() { return g_ready ? g_device : -1; }

// Host-waits for every side stream that holds unjoined work: nothing of it is in flight afterwards, so what was freed on it is reusable and
// the stream is clean.  g_mu held by the caller.
static void drain_side_streams_locked()
{
    for (int k = 0; k < IOPX_SIDE_STREAMS; ++k) {
        if (!g_side_dirty[k]) continue;
        (void)hipStreamSynchronize(g_side_stream[k]);
        g_side_dirty[k] = false;
        std::lock_guard<std::mutex> lt(g_tmp_mu);
        g_tmp_free.insert(g_tmp_free.end(), g_tmp_quarantine[k].begin(), g_tmp_quarantine[k].end());
        g_tmp_quarantine[k].clear();
    }
}

// own = true: back to the library's private stream; otherwise the caller's handle as given, 0 included (the legacy default
// stream, which is what torch's default stream is)
int set_stream(void *s, bool own)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    // temporaries are recycled in stream order: drain the old stream before work moves to another one.  Read, drain and
    // switch under the lock, so a concurrent iopx_set_stream / iopx_use_own_stream from another host thread cannot make this
    // call drain a stream that is no longer (or not yet) the current one
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cur_side >= 0) return fail(IOPX_ERR_LOGIC, "iopx_set_stream inside a side-stream section");
    drain_side_streams_locked();
    (void)hipStreamSynchronize(g_stream);
    g_caller_stream = !own;
    g_stream = own ? g_own_stream : (hipStream_t)s;
    return IOPX_OK;
}

// ---- cached device temporaries: allocation (the lists are declared with the streams above) ----------

void *tmp_alloc(size_t bytes, size_t *cap)
{
    {
        std::lock_guard<std::mutex> lk(g_tmp_mu);
        int best = -1;
        for (size_t i = 0; i < g_tmp_free.size(); ++i) {
            const size_t c = g_tmp_free[i].cap;
            if (c >= bytes && c <= 2 * bytes + (1u << 20) && (best < 0 || c < g_tmp_free[best].cap)) best = (int)i;
        }
        if (best >= 0) {
            TmpBlock b = g_tmp_free[best];
            g_tmp_free.erase(g_tmp_free.begin() + best);
            *cap = b.cap;
            return b.p;
        }
    }
    void *p = nullptr;
    const size_t c = (bytes + 255) & ~(size_t)255;
    ColdScope cold_("hipMalloc (pool growth)");
    if (hipMalloc(&p, c) != hipSuccess) {
        // out of memory: drop the cache (after draining the streams: quarantined blocks become droppable too) and retry once
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lm(g_mu);
            if (g_cur_side < 0) drain_side_streams_locked();
            else (void)hipStreamSynchronize(g_side_stream[g_cur_side]);      // inside a section: its frees stay quarantined until the join
            (void)hipStreamSynchronize(g_stream);
        }
        std::lock_guard<std::mutex> lk(g_tmp_mu);
        for (auto &b : g_tmp_free) (void)hipFree(b.p);
        g_tmp_free.clear();
        if (hipMalloc(&p, c) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    *cap = c;
    return p;
}

void tmp_free(void *p, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_tmp_mu);
    (g_cur_side >= 0 ? g_tmp_quarantine[g_cur_side] : g_tmp_free).push_back({p, cap});
}

int side_stream_fork(int k)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (k < 0 || k >= IOPX_SIDE_STREAMS) return fail(IOPX_ERR_INVALID_ARGUMENT, "side stream %d", k);
    if (g_cur_side >= 0) return fail(IOPX_ERR_LOGIC, "side_stream_fork inside a side-stream section");
    if (!g_side_stream[k]) {                                    // published only when the stream and both events exist
        hipStream_t st = nullptr;
        hipEvent_t ef = nullptr, ej = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ef, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ej, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (ej) (void)hipEventDestroy(ej);
            if (ef) (void)hipEventDestroy(ef);
            if (st) (void)hipStreamDestroy(st);
            return fail(IOPX_ERR_RUNTIME, "side stream creation failed: %s", hipGetErrorString(e));
        }
        g_fork_event[k] = ef; g_join_event[k] = ej; g_side_stream[k] = st;
    }
    IOPX_HIP(hipEventRecord(g_fork_event[k], g_stream));
    IOPX_HIP(hipStreamWaitEvent(g_side_stream[k], g_fork_event[k], 0));
    g_side_dirty[k] = true;
    return IOPX_OK;
}

// k = -1: the main stream.  A side stream must have been forked since its last join.
int side_stream_select(int k)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (k >= IOPX_SIDE_STREAMS || (k >= 0 && !g_side_dirty[k])) return fail(IOPX_ERR_LOGIC, "side_stream_select(%d) without a fork", k);
    g_cur_side = k < 0 ? -1 : k;
    return IOPX_OK;
}

int side_stream_current() { return g_cur_side; }

int side_stream_join(int k)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (k < 0 || k >= IOPX_SIDE_STREAMS) return fail(IOPX_ERR_INVALID_ARGUMENT, "side stream %d", k);
    if (g_cur_side >= 0) return fail(IOPX_ERR_LOGIC, "side_stream_join inside a side-stream section");
    if (!g_side_dirty[k]) return IOPX_OK;
    IOPX_HIP(hipEventRecord(g_join_event[k], g_side_stream[k]));
    IOPX_HIP(hipStreamWaitEvent(g_stream, g_join_event[k], 0));
    g_side_dirty[k] = false;
    std::lock_guard<std::mutex> lt(g_tmp_mu);
    g_tmp_free.insert(g_tmp_free.end(), g_tmp_quarantine[k].begin(), g_tmp_quarantine[k].end());
    g_tmp_quarantine[k].clear();
    return IOPX_OK;
}

void tmp_trim()
{
    {
        std::lock_guard<std::mutex> lm(g_mu);
        if (g_cur_side < 0) drain_side_streams_locked();
        (void)hipStreamSynchronize(g_stream);
    }
    std::lock_guard<std::mutex> lk(g_tmp_mu);
    for (auto &b : g_tmp_free) (void)hipFree(b.p);
    g_tmp_free.clear();
}

// ---- pinned staging for small uploads -------------------------------------------------------------
struct StageChunk { void *p; size_t cap; hipEvent_t done; bool busy; };
static std::vector<StageChunk> g_stage;
static std::mutex g_stage_mu;

// Per-call constants (shift terms, coefficient lists, pointer tables: a few hundred bytes each, ~140 of them per Aurora proof) are
// delivered by a one-workgroup KERNEL whose argument block carries the bytes: an asynchronous copy makes the queue switch between the
// compute and the copy path and leaves the GPU idle for several microseconds on both sides of it (profiles/r04_gpu_gaps.txt: the idle
// time before the runtime's copy kernels was the largest share of the proof's launch gaps), a kernel launch is one more packet in the
// compute queue.  The launch copies its arguments at enqueue time, so the caller's buffer is free on return, as with the staging path.
#define IOPX_UPLOAD_WORDS 960                    // 3840 bytes of payload: the kernel argument block holds 4 KB
struct UploadBlob { uint32_t w[IOPX_UPLOAD_WORDS]; };
__global__ void k_upload_small(UploadBlob blob, uint32_t *dst, int words, int tail_bytes)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = blob.w[i];
    if (threadIdx.x == 0 && tail_bytes) {
        uint8_t *d = (uint8_t *)(dst + words);
        const uint32_t v = blob.w[words];
        for (int b = 0; b < tail_bytes; ++b) d[b] = (uint8_t)(v >> (8 * b));
    }
}

int upload(void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return IOPX_OK;
    if (bytes <= sizeof(UploadBlob) && ((uintptr_t)dst_dev & 3) == 0) {
        UploadBlob blob;
        memcpy(blob.w, src_host, bytes);
        g_bytes_h2d += bytes;
        { ProfScope ps_("k_upload_small"); hipLaunchKernelGGL(k_upload_small, dim3(1), dim3(256), 0, active_stream(), blob, (uint32_t *)dst_dev, (int)(bytes / 4), (int)(bytes % 4)); }
        IOPX_HIP(hipGetLastError());
        return IOPX_OK;
    }
    std::lock_guard<std::mutex> lk(g_stage_mu);
    StageChunk *c = nullptr;
    for (auto &ch : g_stage) {
        if (ch.busy && hipEventQuery(ch.done) == hipSuccess) ch.busy = false;
        if (!ch.busy && ch.cap >= bytes && (!c || ch.cap < c->cap)) c = &ch;
    }
    if (!c) {
        StageChunk ch;
        ch.cap = bytes < 4096 ? 4096 : bytes;
        ch.busy = false;
        ColdScope cold_("hipHostMalloc");
        hipError_t e = hipHostMalloc(&ch.p, ch.cap, 0);
        if (e != hipSuccess) return fail(IOPX_ERR_RUNTIME, "hipHostMalloc(%zu) failed: %s", ch.cap, hipGetErrorString(e));
        e = hipEventCreateWithFlags(&ch.done, hipEventDisableTiming);
        if (e != hipSuccess) return fail(IOPX_ERR_RUNTIME, "hipEventCreate failed: %s", hipGetErrorString(e));
        g_stage.push_back(ch);
        c = &g_stage.back();
    }
    memcpy(c->p, src_host, bytes);
    IOPX_HIP(copy_h2d(dst_dev, c->p, bytes, active_stream()));
    IOPX_HIP(hipEventRecord(c->done, active_stream()));
    c->busy = true;
    return IOPX_OK;
}

static void *g_bounce = nullptr;
static size_t g_bounce_cap = 0;
// Deferred read-backs.  Between iopx_defer_downloads_begin and _end a deferrable read-back is a small KERNEL that copies its bytes into one device
// arena (a kernel launch is one more packet in the compute queue; an asynchronous copy makes the queue switch to the copy path and back, several
// microseconds of idle GPU on both sides: profiles/r05_gpu_gaps.txt counted 43 of them per proof); _end moves the arena to the host with ONE copy,
// drains the stream once and hands every piece to its destination.  The arena and its pinned twin grow on demand and stay.
struct DeferItem { void *dst; size_t off, bytes; };
static bool g_defer_on = false;
static char *g_defer_dev = nullptr, *g_defer_host = nullptr;
static size_t g_defer_cap = 0, g_defer_used = 0;
static std::vector<DeferItem> g_defer_items;

__global__ void k_copy_small(const uint8_t *src, uint8_t *dst, size_t bytes)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        const size_t words = bytes >> 3;
        for (size_t i = t; i < words; i += step) ((uint64_t *)dst)[i] = ((const uint64_t *)src)[i];
        for (size_t i = (words << 3) + t; i < bytes; i += step) dst[i] = src[i];
    } else {
        for (size_t i = t; i < bytes; i += step) dst[i] = src[i];
    }
}

__global__ void k_copy_d2d(const uint8_t *src, uint8_t *dst, size_t bytes)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const size_t quads = bytes >> 4;
        const uint4 *s4 = (const uint4 *)src;
        uint4 *d4 = (uint4 *)dst;
        for (size_t i = t; i < quads; i += step) d4[i] = s4[i];
        for (size_t i = (quads << 4) + t; i < bytes; i += step) dst[i] = src[i];
    } else if ((((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        const size_t words = bytes >> 3;
        for (size_t i = t; i < words; i += step) ((uint64_t *)dst)[i] = ((const uint64_t *)src)[i];
        for (size_t i = (words << 3) + t; i < bytes; i += step) dst[i] = src[i];
    } else {
        for (size_t i = t; i < bytes; i += step) dst[i] = src[i];
    }
}

__global__ void k_fill_bytes(uint8_t *dst, uint8_t value, size_t bytes)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if (((uintptr_t)dst & 15) == 0) {
        const uint32_t w = 0x01010101u * value;
        const uint4 v = make_uint4(w, w, w, w);
        const size_t quads = bytes >> 4;
        for (size_t i = t; i < quads; i += step) ((uint4 *)dst)[i] = v;
        for (size_t i = (quads << 4) + t; i < bytes; i += step) dst[i] = value;
    } else {
        for (size_t i = t; i < bytes; i += step) dst[i] = value;
    }
}

static unsigned copy_grid(size_t bytes)
{
    const size_t blocks = (bytes + 16 * 256 * 4 - 1) / (16 * 256 * 4);          // four 16-byte accesses per thread
    return (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

int copy_d2d(void *dst_dev, const void *src_dev, size_t bytes)
{
    if (bytes == 0 || dst_dev == src_dev) return IOPX_OK;
    { ProfScope ps_("k_copy_d2d", 2 * bytes); hipLaunchKernelGGL(k_copy_d2d, dim3(copy_grid(bytes)), dim3(256), 0, active_stream(), (const uint8_t *)src_dev, (uint8_t *)dst_dev, bytes); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int fill_bytes(void *dst_dev, int value, size_t bytes)
{
    if (bytes == 0) return IOPX_OK;
    { ProfScope ps_("k_fill_bytes", bytes); hipLaunchKernelGGL(k_fill_bytes, dim3(copy_grid(bytes)), dim3(256), 0, active_stream(), (uint8_t *)dst_dev, (uint8_t)value, bytes); }
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// g_stage_mu held.  Drains the arena's pending pieces first when it has to grow (their bytes live in the old arena).
static int defer_reserve(size_t need)
{
    if (g_defer_used + need <= g_defer_cap) return IOPX_OK;
    if (!g_defer_items.empty()) return IOPX_ERR_RUNTIME;         // caller falls back to an immediate read-back for this piece
    ColdScope cold_("deferred read-back arena");
    size_t cap = g_defer_cap ? g_defer_cap : ((size_t)1 << 20);
    while (cap < need) cap <<= 1;
    if (g_defer_dev) (void)hipFree(g_defer_dev);
    if (g_defer_host) (void)hipHostFree(g_defer_host);
    g_defer_dev = g_defer_host = nullptr; g_defer_cap = 0;
    hipError_t e = hipMalloc((void **)&g_defer_dev, cap);
    if (e == hipSuccess) e = hipHostMalloc((void **)&g_defer_host, cap, 0);
    if (e != hipSuccess) return fail(IOPX_ERR_RUNTIME, "deferred read-back arena of %zu bytes: %s", cap, hipGetErrorString(e));
    g_defer_cap = cap;
    return IOPX_OK;
}

int download(void *dst_host, const void *src_dev, size_t bytes, bool deferrable)
{
    if (bytes == 0) return IOPX_OK;
    std::unique_lock<std::mutex> lk(g_stage_mu);
    if (deferrable && g_defer_on) {
        const size_t need = (bytes + 63) & ~(size_t)63;          // 64-byte slots
        const int rc = defer_reserve(need);
        if (rc == IOPX_OK) {
            const unsigned blocks = (unsigned)std::min<size_t>((bytes + 4095) / 4096, 64);
            { ProfScope ps_("k_copy_small"); hipLaunchKernelGGL(k_copy_small, dim3(blocks ? blocks : 1), dim3(256), 0, active_stream(), (const uint8_t *)src_dev, (uint8_t *)g_defer_dev + g_defer_used, bytes); }
            IOPX_HIP(hipGetLastError());
            g_defer_items.push_back({dst_host, g_defer_used, bytes});
            g_defer_used += need;
            return IOPX_OK;
        }
        if (!g_defer_dev) return rc;                             // the arena could not be created at all
    }
    if (bytes > ((size_t)4 << 20)) {                    // large read-backs (test helpers): the plain path
        lk.unlock();
        IOPX_HIP(copy_d2h(dst_host, src_dev, bytes, active_stream()));
        IOPX_HIP(hipStreamSynchronize(active_stream()));
        return IOPX_OK;
    }
    if (bytes > g_bounce_cap) {
        if (g_bounce) (void)hipHostFree(g_bounce);
        g_bounce = nullptr; g_bounce_cap = 0;
        const size_t cap = bytes < 65536 ? 65536 : bytes;
        ColdScope cold_("hipHostMalloc");
        hipError_t e = hipHostMalloc(&g_bounce, cap, 0);
        if (e != hipSuccess) return fail(IOPX_ERR_RUNTIME, "hipHostMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        g_bounce_cap = cap;
    }
    IOPX_HIP(copy_d2h(g_bounce, src_dev, bytes, active_stream()));
    IOPX_HIP(hipStreamSynchronize(active_stream()));
    memcpy(dst_host, g_bounce, bytes);
    return IOPX_OK;
}

int defer_downloads_begin()
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    if (g_defer_on) return fail(IOPX_ERR_LOGIC, "iopx_defer_downloads_begin: already deferring");
    g_defer_on = true;
    g_defer_used = 0;
    return IOPX_OK;
}

int defer_downloads_end()
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    if (!g_defer_on) return fail(IOPX_ERR_LOGIC, "iopx_defer_downloads_end without begin");
    g_defer_on = false;
    // a deferrable read-back may have been queued inside a side-stream section: the main stream waits for the side streams, draining it covers both
    for (int k = 0; k < IOPX_SIDE_STREAMS; ++k) if (side_stream_current() < 0) (void)side_stream_join(k);
    hipError_t e = hipSuccess;
    if (g_defer_used) e = copy_d2h(g_defer_host, g_defer_dev, g_defer_used, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e == hipSuccess) for (auto &it : g_defer_items) memcpy(it.dst, g_defer_host + it.off, it.bytes);
    g_defer_items.clear();
    g_defer_used = 0;
    IOPX_HIP(e);
    return IOPX_OK;
}

// ---- domain-dependent tables kept on the device ----------------------------------------------------
static std::mutex g_dtab_mu;
static std::map<std::vector<uint64_t>, std::shared_ptr<DevBuf>> g_dtabs;

int cached_domain_table(const std::vector<uint64_t> &key, const std::function<int(std::vector<uint64_t> &)> &build, TmpBuf &out)
{
    std::lock_guard<std::mutex> lk(g_dtab_mu);
    auto it = g_dtabs.find(key);
    if (it == g_dtabs.end()) {
        ColdScope cold_("domain table");
        std::vector<uint64_t> words;
        int rc = build(words);
        if (rc != IOPX_OK) return rc;
        std::shared_ptr<DevBuf> buf(new DevBuf());
        if ((rc = buf->alloc(words.size() * 8)) != IOPX_OK) return rc;
        if ((rc = upload(buf->p, words.data(), words.size() * 8)) != IOPX_OK) return rc;
        if (g_dtabs.size() >= 64) g_dtabs.clear();          // borrowers keep their entries alive
        it = g_dtabs.emplace(key, buf).first;
    }
    out.borrow(it->second->p, it->second->bytes, it->second);
    return IOPX_OK;
}

void clear_domain_tables()
{
    std::lock_guard<std::mutex> lk(g_dtab_mu);
    g_dtabs.clear();
}

// ---- options ----------------------------------------------------------------------------------------
static std::mutex g_opt_mu;
static std::map<std::string, std::pair<bool, int>> g_opts;           // name -> (has a value, value); an entry without a value: the environment was asked and had none
int opt(const char *name, int dflt)
{
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_opts.find(name);
    if (it == g_opts.end()) {
        const char *v = getenv(name);                                  // the library's only read of the environment
        it = g_opts.emplace(name, (v && *v) ? std::make_pair(true, atoi(v)) : std::make_pair(false, 0)).first;
    }
    return it->second.first ? it->second.second : dflt;
}
int opt_range(const char *name, int dflt, int lo, int hi)
{
    const int x = opt(name, dflt);
    return x < lo ? lo : (x > hi ? hi : x);
}

// ---- one-time costs, by label ---------------------------------------------------------------------
static std::mutex g_cold_mu;
static std::map<std::string, std::pair<size_t, double>> g_cold;       // label -> (count, ms)
static long long now_ns() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; }
void cold_add(const char *label, double ms)
{
    std::lock_guard<std::mutex> lk(g_cold_mu);
    auto &e = g_cold[label];
    e.first += 1;
    e.second += ms;
}
ColdScope::ColdScope(const char *l) : label(l), t0_ns(now_ns()) {}
ColdScope::~ColdScope() { cold_add(label, (double)(now_ns() - t0_ns) * 1e-6); }

// ---- per-kernel profiling -----------------------------------------------------------------------
struct ProfRec { const char *name; hipEvent_t a, b; size_t bytes, products; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(const char *name, size_t work_bytes, size_t work_products) : slot(-1)
{
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name;
    r.bytes = work_bytes;
    r.products = work_products;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, active_stream());
    slot = (int)g_prof.size();
    g_prof.push_back(r);
}

ProfScope::~ProfScope()
{
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, active_stream());
}

} // namespace iopx

extern "C" {

int iopx_profile_begin(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    for (auto &r : iopx::g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    iopx::g_prof.clear();
    iopx::g_prof_on = true;
    return IOPX_OK;
}

// Stops profiling and writes one line per kernel name: "<name> <launches> <total_ms> <algorithmic_bytes> <field_products>\n".
int iopx_profile_report(char *buf, size_t cap)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    iopx::g_prof_on = false;
    for (int k = 0; k < IOPX_SIDE_STREAMS; ++k) if (iopx::side_stream_current() < 0) (void)iopx::side_stream_join(k);     // events recorded on a side stream before the profile began
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    struct Agg { size_t launches = 0; double ms = 0; double bytes = 0; double products = 0; };
    std::map<std::string, Agg> agg;
    for (auto &r : iopx::g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &e = agg[r.name];
            e.launches += 1;
            e.ms += ms;
            e.bytes += (double)r.bytes;
            e.products += (double)r.products;
        }
    }
    // idle time between consecutive profiled launches (end event of one to start event of the next): where host work, read-backs and launch
    // latency are exposed.  "@idle <total ms> <gaps>" and the twelve longest as "@gap <us> <after> <before>"
    std::string gaps_out;
    {
        std::vector<std::pair<float, size_t>> gaps;
        double idle = 0;
        for (size_t i = 0; i + 1 < iopx::g_prof.size(); ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, iopx::g_prof[i].b, iopx::g_prof[i + 1].a) == hipSuccess && ms > 0) { gaps.emplace_back(ms, i); idle += ms; }
        }
        std::sort(gaps.begin(), gaps.end(), [](const std::pair<float, size_t> &x, const std::pair<float, size_t> &y) { return x.first > y.first; });
        char line[256];
        snprintf(line, sizeof(line), "@idle %.6f %zu\n", idle, gaps.size());
        gaps_out += line;
        for (size_t k = 0; k < gaps.size() && k < 12; ++k) {
            snprintf(line, sizeof(line), "@gap %.1f %s %s\n", gaps[k].first * 1e3, iopx::g_prof[gaps[k].second].name, iopx::g_prof[gaps[k].second + 1].name);
            gaps_out += line;
        }
    }
    for (auto &r : iopx::g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    iopx::g_prof.clear();
    std::string out;
    for (auto &kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s %zu %.6f %.0f %.0f\n", kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.bytes, kv.second.products);
        out += line;
    }
    out += gaps_out;
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return IOPX_OK;
}

// "<label with spaces replaced by _> <count> <total ms>\n" per label of one-time host-side cost since the last reset
int iopx_cold_stats(char *buf, size_t cap, int reset)
{
    std::lock_guard<std::mutex> lk(iopx::g_cold_mu);
    std::string out;
    for (auto &kv : iopx::g_cold) {
        std::string l = kv.first;
        for (char &ch : l) if (ch == ' ') ch = '_';
        char line[256];
        snprintf(line, sizeof(line), "%s %zu %.4f\n", l.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    if (reset) iopx::g_cold.clear();
    return IOPX_OK;
}
int iopx_cold_add(const char *label, double ms) { if (label) iopx::cold_add(label, ms); return IOPX_OK; }

int iopx_set_option(const char *name, int value)
{
    if (!name || !*name) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_set_option: empty name");
    std::lock_guard<std::mutex> lk(iopx::g_opt_mu);
    iopx::g_opts[name] = std::make_pair(true, value);
    return IOPX_OK;
}
int iopx_clear_option(const char *name)          // back to the environment's value (asked again at the next lookup) or the default
{
    if (!name) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_clear_option: null name");
    std::lock_guard<std::mutex> lk(iopx::g_opt_mu);
    iopx::g_opts.erase(name);
    return IOPX_OK;
}
int iopx_get_option(const char *name, int dflt) { return name ? iopx::opt(name, dflt) : dflt; }

int iopx_version(void) { return 100; }

const char *iopx_last_error(void) { return iopx::last_error(); }

int iopx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int iopx_init(int device)
{
    int n = iopx_device_count();
    if (n <= 0) return iopx::fail(IOPX_ERR_NO_DEVICE, "no HIP device available; libiop_amd has no CPU fallback");
    if (device < 0 || device >= n) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, n);
    // the stream, the cached temporaries, the staging events and every plan belong to the first device used
    const int bound = iopx::bound_device();
    if (bound >= 0 && bound != device)
        return iopx::fail(IOPX_ERR_LOGIC, "libiop_amd is bound to device %d (one process per GPU); cannot rebind to device %d", bound, device);
    IOPX_HIP(hipSetDevice(device));
    return iopx::ensure_device();
}

int iopx_set_stream(void *hip_stream) { return iopx::set_stream(hip_stream, false); }
int iopx_use_own_stream(void) { return iopx::set_stream(nullptr, true); }

int iopx_synchronize(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    for (int k = 0; k < IOPX_SIDE_STREAMS; ++k) // the main stream waits for the side stream; draining it then covers both
        if ((rc = iopx::side_stream_join(k)) != IOPX_OK) return rc;
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    return IOPX_OK;
}

int iopx_side_stream_begin(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    // while the HIP-event profiler records, the section stays on the main stream: every kernel is then timed running alone and the
    // durations add up (what bench.py's per-kernel figures and roofline.achieved are computed from)
    if (iopx::g_prof_on) {
        if (iopx::g_side_elided || iopx::side_stream_current() >= 0) return iopx::fail(IOPX_ERR_LOGIC, "side_stream_begin inside a side-stream section");
        iopx::g_side_elided = true;
        return IOPX_OK;
    }
    if ((rc = iopx::side_stream_fork(0)) != IOPX_OK) return rc;
    return iopx::side_stream_select(0);
}

int iopx_side_stream_end(void)
{
    if (iopx::g_side_elided) { iopx::g_side_elided = false; return IOPX_OK; }
    if (iopx::side_stream_current() != 0) return iopx::fail(IOPX_ERR_LOGIC, "iopx_side_stream_end without begin");
    return iopx::side_stream_select(-1);
}

int iopx_side_stream_join(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    return iopx::side_stream_join(0);
}

int iopx_malloc(void **dptr, size_t bytes)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!dptr) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_malloc: null out pointer");
    IOPX_HIP(hipMalloc(dptr, bytes ? bytes : 8));
    return IOPX_OK;
}

int iopx_free(void *dptr)
{
    if (!dptr) return IOPX_OK;
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(hipFree(dptr));
    return IOPX_OK;
}

int iopx_defer_downloads_begin(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    return iopx::defer_downloads_begin();
}

int iopx_defer_downloads_end(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    return iopx::defer_downloads_end();
}

int iopx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(iopx::copy_h2d(dst_dev, src_host, bytes, iopx::stream()));
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    return IOPX_OK;
}

int iopx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    return iopx::download(dst_host, src_dev, bytes);
}

// the same read-back, queued (not waited for) between iopx_defer_downloads_begin and _end; immediate outside such a window
int iopx_memcpy_d2h_deferrable(void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    return iopx::download(dst_host, src_dev, bytes, /*deferrable=*/true);
}

} // extern "C"
