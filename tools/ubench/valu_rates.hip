// Micro-benchmark: issue rate of the VALU ops the GF(2^192) multiply is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 4096
template<int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed + 77, sc = 0;
    uint32_t d2[4] = { seed, 0x3ff00000u | (seed & 0xfffff), seed * 3u, 0x3fe00000u | (seed & 0xffff) };
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_bfe_i32 %0, %0, 5, 1" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(b), "v"(c) : "vcc");
            if (OP == 9) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(a[i]) : "v"(b), "s"(seed));
            if (OP == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 12) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 13) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(*(uint64_t *)&a[(i + 2) & 6]));
            if (OP == 14) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i & 6]), "+v"(b), "+v"(a[(i & 6) + 1]), "+v"(c) : : "vcc");
            if (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(*(uint64_t *)&a[i & 6]));
            if (OP == 17) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 18) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(a[i]) : "v"(b), "s"(seed));
            if (OP == 19) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 20) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xe4" : "+v"(a[i]) : "s"(seed), "v"(c));
            if (OP == 21) asm volatile("v_readfirstlane_b32 %1, %0\n\tv_xor_b32 %0, %1, %0" : "+v"(a[i]), "=s"(sc) : );
            if (OP == 22) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            if (OP == 23) asm volatile("v_and_b32 %0, 0x55555555, %0" : "+v"(a[i]));
            // round 5: the double-precision pipe (a 181-bit Montgomery product on FMA limbs) and the other integer multiplies
            if (OP == 24) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*(double *)&a[i & 6]) : "v"(*(double *)&d2[0]), "v"(*(double *)&d2[2]));
            if (OP == 25) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *)&a[i & 6]) : "v"(*(double *)&d2[0]));
            if (OP == 26) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&a[i & 6]) : "v"(*(double *)&d2[0]));
            if (OP == 27) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 28) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 29) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(*(double *)&a[i & 6]) : "v"(b));
            if (OP == 30) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(b), "v"(c) : "vcc");
            if (OP == 31) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(*(uint64_t *)&d2[0]), "v"(*(uint64_t *)&d2[2]));
            if (OP == 32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int OP> float run(const char *name, uint32_t *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;          // 8 blocks of 4 waves per CU: 8 waves/SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * REP * 8;
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, ops / ms / 1e9);
    return ms;
}

int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32 (VOP2)", d);
    run<4>("v_and_b32 (VOP2)", d);
    run<5>("v_lshlrev_b32 (VOP2)", d);
    run<1>("v_bitop3_b32 (3 vgpr)", d);
    run<9>("v_bitop3_b32 (2 vgpr+sgpr)", d);
    run<6>("v_xor3_b32", d);
    run<10>("v_and_or_b32", d);
    run<2>("v_alignbit_b32", d);
    run<3>("v_bfe_i32", d);
    run<12>("v_perm_b32", d);
    run<11>("v_mul_u32_u24", d);
    run<7>("v_mul_lo_u32", d);
    run<8>("v_mad_u64_u32", d);
    run<13>("v_lshl_add_u64 (64-bit add)", d);
    run<14>("v_add_co + v_addc_co pair", d);
    run<15>("v_add_u32", d);
    run<17>("v_add3_u32", d);
    run<16>("v_lshrrev_b64", d);
    run<18>("v_bitop3 0xe4 (vgpr, vgpr, sgpr)", d);
    run<19>("v_bitop3 0xe4 (3 vgpr)", d);
    run<20>("v_bitop3 0xe4 (sgpr, vgpr, vgpr)", d);
    run<22>("v_xor_b32 (VOP2, sgpr src0)", d);
    run<23>("v_and_b32 (VOP2, literal)", d);
    run<21>("v_readfirstlane + v_xor pair", d);
    run<24>("v_fma_f64", d);
    run<25>("v_add_f64", d);
    run<26>("v_mul_f64", d);
    run<29>("v_cvt_f64_u32", d);
    run<27>("v_mul_hi_u32", d);
    run<28>("v_mad_u32_u24", d);
    run<30>("v_mad_i64_i32", d);
    run<32>("v_fma_f32", d);
    run<31>("v_pk_fma_f32 (2 lanes of f32 per op)", d);
    return 0;
}
