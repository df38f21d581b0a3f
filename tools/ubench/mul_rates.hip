// Micro-benchmark: in-register throughput of the GF(2^192) products (no memory traffic in the loop) against the number of wavefronts per
// SIMD.  A workgroup is 256 threads (one wavefront per SIMD of its CU) and asks for 160 KB / k of LDS, so exactly k workgroups — k waves per
// SIMD — are resident, whatever the kernel's register count allows beyond that.  Round 5: general = gf_mul (114 VGPRs: at most 4 waves),
// lean = gf_mul_lean (53 VGPRs), uniform = the comb product, uniform2 = the comb product
// serving two operands per window dispatch (tools/ubench/gen_comb_dual.py: a micro-benchmark, not in the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I libiop_amd/csrc/include -mllvm -pragma-unroll-threshold=1000000 tools/ubench/mul_rates.hip -o tools/ubench/mul_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../libiop_amd/csrc/gf192_dev.h"
#include "comb_dual.h"

// two products by one wave-uniform multiplier with one window dispatch (tools/ubench/gen_comb_dual.py)
__device__ __forceinline__ void gf_mul_uniform2(gf192 &a, gf192 &e, const gf192 &c_uniform)
{
    uint32_t c[6], r[12], q[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) { c[i] = __builtin_amdgcn_readfirstlane(c_uniform.w[i]); r[i] = a.w[i]; q[i] = e.w[i]; }
    comb_clmul_192_uniform2(r, q, c);
    a = gf_reduce(r);
    e = gf_reduce(q);
}

#define CHAIN 64
template<int MODE>
__global__ void __launch_bounds__(256) k(const uint64_t *in, uint64_t *out)
{
    extern __shared__ uint64_t lds[];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gf192 x = gf_load(in, i), y = gf_load(in, (i * 7 + 3) & 0xffff);
    const gf192 u = gf_load(in, 5);
    if (in == out) lds[threadIdx.x] = x.w[0];            // never true: keeps the allocation
    for (int r = 0; r < (MODE == 3 ? CHAIN / 2 : CHAIN); ++r) {
        if (MODE == 3) gf_mul_uniform2(x, y, u);
        else if (MODE == 0) x = gf_mul(x, y);
        else if (MODE == 2) x = gf_mul_lean(x, y);
        else x = gf_mul_uniform(x, u);
    }
    if (MODE == 3) gf_add_to(x, y);
    gf_store(out, i, x);
}

__global__ void k_check(const uint64_t *in, unsigned *bad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gf192 x = gf_load(in, i), y = gf_load(in, (i * 7 + 3) & 0xffff);
    const gf192 u = gf_load(in, 5);
    const gf192 rx = gf_mul_uniform(x, u), ry = gf_mul_uniform(y, u), gx = gf_mul(x, u);
    gf_mul_uniform2(x, y, u);
    bool ok = true;
    for (int w = 0; w < 6; ++w) ok = ok && x.w[w] == rx.w[w] && y.w[w] == ry.w[w] && gx.w[w] == rx.w[w];
    if (!ok) atomicAdd(bad, 1u);
}

template<int MODE> void run(const char *name, const uint64_t *in, uint64_t *out, int waves)
{
    const int blocks = 256 * waves * 8;
    const size_t lds = (160 * 1024) / waves - 256;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, in, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, in, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rate = (double)blocks * 256 * CHAIN / ms * 1e3;
    printf("%-8s %d waves/SIMD (asked)  %8.3f ms  %.3e products/s  %.0f cycles per wave-product per SIMD\n", name, waves, ms, rate, 1024.0 * 2.4e9 * 64 / rate);
}

int main()
{
    uint64_t *in, *out;
    const size_t n = (size_t)256 * 8 * 8 * 256;
    hipMalloc(&in, n * 24); hipMalloc(&out, n * 24);
    {
        std::vector<uint64_t> h(n * 3);
        uint64_t z = 0x9e3779b97f4a7c15ull;
        for (auto &v : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = z; }
        hipMemcpy(in, h.data(), n * 24, hipMemcpyHostToDevice);
        unsigned *bad, hb = 0;
        hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, 0, in, bad);
        hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("uniform2 against uniform and general on 65536 random operand pairs: %u mismatches\n", hb);
    }
    for (int w : {2, 3, 4, 5, 6, 8}) { run<0>("general", in, out, w); run<2>("lean", in, out, w); run<1>("uniform", in, out, w); run<3>("uniform2", in, out, w); }
    return 0;
}
