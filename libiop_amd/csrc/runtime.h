// Internal runtime of libiop_amd: error reporting, the current HIP stream, small helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
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

// Per-kernel timing with HIP events on the library's stream (iopx_profile_begin / iopx_profile_report).
// Costs nothing when profiling is off.  Usage: { ProfScope ps("k_name"); hipLaunchKernelGGL(...); }
struct ProfScope {
    int slot;
    explicit ProfScope(const char *name);
    ~ProfScope();
};

static inline size_t ceil_log2(size_t n)
{
    size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) { n >>= 1; ++r; }
    return r;
}

} // namespace iopx
