// TEST INFRASTRUCTURE ONLY — a single-thread-per-workgroup stand-in for the HIP runtime.
//
// There is no GPU in the development container, so the CPU test-suite compiles the PRODUCT kernel
// sources (libiop_amd/csrc/*.hip, unmodified, no #ifdefs in them) with g++ against this header and runs
// every workgroup sequentially with blockDim = 1.  All kernels are written as block-size-agnostic
// strided loops separated by __syncthreads(), so one thread per workgroup executes the same index
// arithmetic, tile schedules and field arithmetic as the GPU does.  This validates kernel LOGIC before
// GPU time is spent; it is not shipped, not loaded by libiop_amd, and not a fallback (the product
// library links the real HIP runtime and fails without a device).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v = { x, y, z, w }; return v; }

static inline void __syncthreads() {}
// one thread at a time: a ballot sees the calling lane only
static inline unsigned long long __ballot(int pred) { return pred ? ~0ull : 0ull; }      // (the emulated thread stands for every lane of its wavefront)
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline void __threadfence_block() {}
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v)
{
    const unsigned long long old = *p;
    if (v < old) *p = v;
    return old;
}
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v)
{
    const unsigned long long old = *p;
    *p = old + v;
    return old;
}
static inline void __threadfence() {}
static inline uint32_t __brev(uint32_t x)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// gfx950 builtins used by the kernels, bit-exact software models
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, uint32_t tt)
{
    if (tt == 0x78) return a ^ (b & c);      // fast path for the form the field multiply uses
    if (tt == 0x96) return a ^ b ^ c;
    if (tt == 0xE4) return (a & c) | (b & ~c);
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) {
        const uint32_t idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1);
        r |= ((tt >> idx) & 1u) << i;
    }
    return r;
}
static inline int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width)
{
    const uint32_t x = ((uint32_t)v >> off) & ((width >= 32) ? 0xFFFFFFFFu : ((1u << width) - 1));
    const uint32_t sign = 1u << (width - 1);
    return (int)((x ^ sign) - sign);
}

static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t x) { return x; }
// v_alignbit_b32: the low 32 bits of ((hi:lo) >> (s & 31))
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31));
}

static inline unsigned long long __brevll(unsigned long long x)
{
    unsigned long long r = 0;
    for (int i = 0; i < 64; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 8); return *p ? hipSuccess : hipErrorUnknown; }
template<typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
static inline hipError_t hipFreeAsync(void *p, hipStream_t) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

typedef void *hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 8); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// LDS: one static buffer, the size of a CU's LDS
namespace iopx { extern uint64_t iopx_smem[]; }
void emu_launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emu_launch((grid), (block), (lds), [&]() { kernel(__VA_ARGS__); })
