// Micro-benchmark: in-register throughput of the GF(2^192) products (no memory traffic in the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../libiop_amd/csrc/gf192_dev.h"

#define CHAIN 64
template<int MODE>
__global__ void __launch_bounds__(256) k(const uint64_t *in, uint64_t *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gf192 x = gf_load(in, i), y = gf_load(in, (i * 7 + 3) & 0xffff);
    const gf192 u = gf_load(in, 5);
    for (int r = 0; r < CHAIN; ++r) {
        if (MODE == 0) x = gf_mul(x, y);
        else x = gf_mul_uniform(x, u);
    }
    gf_store(out, i, x);
}

template<int MODE> void run(const char *name, const uint64_t *in, uint64_t *out, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s blocks=%5d  %8.3f ms  %.3e mult/s\n", name, blocks, ms, (double)blocks * 256 * CHAIN / ms * 1e3);
}

int main()
{
    const int maxb = 256 * 16;
    uint64_t *in, *out;
    hipMalloc(&in, (size_t)maxb * 256 * 24); hipMalloc(&out, (size_t)maxb * 256 * 24);
    hipMemset(in, 0x5a, (size_t)maxb * 256 * 24);
    for (int b : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) { run<0>("general", in, out, b); run<1>("uniform", in, out, b); }
    return 0;
}
