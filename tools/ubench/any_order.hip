// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  Two launches of a kernel that occupies half of the chip's workgroup slots
// for ~1 ms each: back to back in order they take 2 x, overlapped 1 x.  (hip_ext.h notes the flag as unsupported on GFX9 for the module-launch form.)
// Build: hipcc --offload-arch=gfx950 -O3 -o any_order any_order.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ void spin(uint64_t *out, long long cycles)
{
    const long long t0 = wall_clock64();
    uint64_t x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42) out[0] = x;
}

static float run(int mode, hipStream_t s, uint64_t *d, int wgs, long long cyc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 2; ++i) {
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, d, cyc);
        else hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, nullptr, nullptr, i == 0 ? 0u : (unsigned)hipExtAnyOrderLaunch, d, cyc);
    }
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    uint64_t *d;
    hipMalloc(&d, 64);
    const long long cyc = 100000000 / 1000;        // clock64 ticks at 100 MHz: 1 ms
    for (int rep = 0; rep < 3; ++rep) {
        const float a = run(0, s, d, 256, cyc), b = run(1, s, d, 256, cyc);
        printf("two 256-workgroup launches of ~1 ms: in order %.3f ms, second with hipExtAnyOrderLaunch %.3f ms\n", a, b);
    }
    return 0;
}
