// libiop_amd runtime: device binding, stream, memory helpers, error strings (see include/libiop_amd.h).
#include "runtime.h"
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace iopx {

static thread_local char g_err[512] = "";
static hipStream_t g_stream = nullptr;      // stream in use
static hipStream_t g_own_stream = nullptr;  // created lazily
static bool g_ready = false;
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

hipStream_t stream() { return g_stream; }

int ensure_device()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready) return IOPX_OK;
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
    if (!g_stream) g_stream = g_own_stream;
    g_ready = true;
    return IOPX_OK;
}

int set_stream(void *s)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    g_stream = s ? (hipStream_t)s : g_own_stream;
    return IOPX_OK;
}

// ---- per-kernel profiling -----------------------------------------------------------------------
struct ProfRec { const char *name; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(const char *name) : slot(-1)
{
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, g_stream);
    slot = (int)g_prof.size();
    g_prof.push_back(r);
}

ProfScope::~ProfScope()
{
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, g_stream);
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

// Stops profiling and writes one line per kernel name: "<name> <launches> <total_ms>\n".
int iopx_profile_report(char *buf, size_t cap)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    iopx::g_prof_on = false;
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    std::map<std::string, std::pair<size_t, double>> agg;
    for (auto &r : iopx::g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &e = agg[r.name];
            e.first += 1;
            e.second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    iopx::g_prof.clear();
    std::string out;
    for (auto &kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s %zu %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return IOPX_OK;
}

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
    IOPX_HIP(hipSetDevice(device));
    return iopx::ensure_device();
}

int iopx_set_stream(void *hip_stream) { return iopx::set_stream(hip_stream); }

int iopx_synchronize(void)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    return IOPX_OK;
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

int iopx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, iopx::stream()));
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    return IOPX_OK;
}

int iopx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    IOPX_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, iopx::stream()));
    IOPX_HIP(hipStreamSynchronize(iopx::stream()));
    return IOPX_OK;
}

} // extern "C"
