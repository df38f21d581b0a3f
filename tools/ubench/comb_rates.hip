// Micro-benchmark (MI355X): (1) issue rate of the instruction patterns the wave-uniform comb product is made of,
// (2) whole-product rate of every candidate schedule in comb_variants.h, each at 1..8 waves per SIMD and checked
// against the general product.  Build: python3 gen_comb_variants.py && hipcc --offload-arch=gfx950 -O3 -I../../libiop_amd/csrc/include comb_rates.hip -o comb_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../libiop_amd/csrc/gf192_dev.h"
#include "comb_variants.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------------------------
// (1) issue patterns: one `round` = 6 windows of 7 relative v_xor (+ the SALU that precedes each window in pattern P)
//     + (patterns >= 10) the 12-instruction accumulator shift.  Table window v[16:37], accumulator v[4:15].
// ------------------------------------------------------------------------------------------------------------------
#define ROUNDS 512
template<int P>
__global__ void __launch_bounds__(256) k_issue(uint32_t *out, uint32_t seed)
{
    extern __shared__ uint32_t dummy_lds[];
    uint32_t cnt, st, sm0, acc = threadIdx.x;
    uint32_t c0 = seed * 0x9e3779b9u, c1 = seed * 0x85ebca6bu;
    c0 = __builtin_amdgcn_readfirstlane(c0) & 0x0f0f0f0fu; c1 = __builtin_amdgcn_readfirstlane(c1);
#define W7(pre) pre \
    "v_xor_b32 v4, v4, v16\n\tv_xor_b32 v5, v5, v17\n\tv_xor_b32 v6, v6, v18\n\tv_xor_b32 v7, v7, v19\n\t" \
    "v_xor_b32 v8, v8, v20\n\tv_xor_b32 v9, v9, v21\n\tv_xor_b32 v10, v10, v22\n\t"
#define SH12 "v_alignbit_b32 v15, v15, v14, 28\n\tv_alignbit_b32 v14, v14, v13, 28\n\tv_alignbit_b32 v13, v13, v12, 28\n\t" \
    "v_alignbit_b32 v12, v12, v11, 28\n\tv_alignbit_b32 v11, v11, v10, 28\n\tv_alignbit_b32 v10, v10, v9, 28\n\t" \
    "v_alignbit_b32 v9, v9, v8, 28\n\tv_alignbit_b32 v8, v8, v7, 28\n\tv_alignbit_b32 v7, v7, v6, 28\n\t" \
    "v_alignbit_b32 v6, v6, v5, 28\n\tv_alignbit_b32 v5, v5, v4, 28\n\tv_lshlrev_b32 v4, 4, v4\n\t"
#define S0 ""
#define S1 "s_set_gpr_idx_idx %[c0]\n\t"
#define S2 "s_bfe_u32 %[st], %[c0], 0x40008\n\ts_set_gpr_idx_idx %[st]\n\t"
#define S3 "s_bfe_u32 %[st], %[c0], 0x40008\n\ts_mul_i32 %[st], %[st], 1\n\ts_set_gpr_idx_idx %[st]\n\t"
#define BODY(S, TAIL) \
    asm volatile("s_mov_b32 %[sm0], m0\n\ts_mov_b32 %[cnt], " #TAIL "\n\t" \
        "v_mov_b32 v4, %[acc]\n\tv_mov_b32 v5, %[acc]\n\tv_mov_b32 v6, %[acc]\n\tv_mov_b32 v7, %[acc]\n\tv_mov_b32 v8, %[acc]\n\tv_mov_b32 v9, %[acc]\n\t" \
        "v_mov_b32 v10, %[acc]\n\tv_mov_b32 v11, %[acc]\n\tv_mov_b32 v12, %[acc]\n\tv_mov_b32 v13, %[acc]\n\tv_mov_b32 v14, %[acc]\n\tv_mov_b32 v15, %[acc]\n\t" \
        "s_mov_b32 %[st], 0\n\ts_set_gpr_idx_on %[st], 2\n\t" \
        "1:\n\t" W7(S) W7(S) W7(S) W7(S) W7(S) W7(S)
#define TAILA "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t" \
        "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[sm0]\n\t" \
        "v_xor_b32 %[acc], v4, v5\n\tv_xor_b32 %[acc], %[acc], v10\n\tv_xor_b32 %[acc], %[acc], v15\n\t" \
        : [acc] "+v"(acc), [cnt] "=&s"(cnt), [st] "=&s"(st), [sm0] "=&s"(sm0) : [c0] "s"(c0), [c1] "s"(c1) \
        : "scc", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37")
    // the table window is only read: its contents do not matter for the rate (index stays < 16 * 7)
    if (P == 0) { BODY(S0, 512) TAILA; }
    if (P == 1) { BODY(S1, 512) TAILA; }
    if (P == 2) { BODY(S2, 512) TAILA; }
    if (P == 3) { BODY(S3, 512) TAILA; }
    if (P == 10) { BODY(S0, 512) "s_set_gpr_idx_idx 0\n\t" SH12 TAILA; }
    if (P == 11) { BODY(S1, 512) "s_set_gpr_idx_idx 0\n\t" SH12 TAILA; }
    if (P == 12) { BODY(S2, 512) "s_set_gpr_idx_idx 0\n\t" SH12 TAILA; }
    if (P == 13) { BODY(S3, 512) "s_set_gpr_idx_idx 0\n\t" SH12 TAILA; }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc + dummy_lds[0] * 0;
}

// plain (non-relative) streams for reference: 42 independent v_xor per round; 12 v_alignbit per round
template<int P>
__global__ void __launch_bounds__(256) k_plain(uint32_t *out, uint32_t seed)
{
    extern __shared__ uint32_t dummy_lds[];
    uint32_t a[12];
    for (int i = 0; i < 12; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    uint32_t b = seed ^ 0x9e3779b9u;
    for (int r = 0; r < ROUNDS; ++r) {
        if (P == 0) {
#pragma unroll
            for (int w = 0; w < 6; ++w)
#pragma unroll
                for (int i = 0; i < 7; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll
                for (int i = 11; i > 0; --i) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(a[i]) : "v"(a[i - 1]));
                asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(a[0]));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 12; ++i) s ^= a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + dummy_lds[0] * 0;
}

// ------------------------------------------------------------------------------------------------------------------
// (2) whole products
// ------------------------------------------------------------------------------------------------------------------
#define CHAIN 48
template<int V>
__device__ __forceinline__ gf192 mulv(const gf192 &a, const gf192 &cu)
{
    uint32_t c[6], r[12];
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_readfirstlane(cu.w[i]);
    if (V == 0) comb_v0(r, a.w, c);
    if (V == 1) comb_v1(r, a.w, c);
    if (V == 2) comb_v2(r, a.w, c);
    if (V == 3) comb_v3(r, a.w, c);
    if (V == 4) comb_v4(r, a.w, c);
    if (V == 5) comb_v3hi(r, a.w, c);
    if (V == 6) comb_j0(r, a.w, c);
    if (V == 7) comb_j1(r, a.w, c);
    if (V == 8) comb_j2(r, a.w, c);
    return gf_reduce(r);
}

template<int V>
__global__ void __launch_bounds__(256) k_mul(const uint64_t *in, uint64_t *out, int check)
{
    extern __shared__ uint32_t dummy_lds[];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gf192 x = gf_load(in, i & 0xffff);
    gf192 u = gf_load(in, 5 + (blockIdx.x & 7));
    if (V == -2) u = gf_load(in, (i * 7 + 3) & 0xffff);       // both operands per lane
    if (check) {
        // one product, compared with the general multiplier (top bits of x set on some lanes to exercise the a_hi split)
        if (threadIdx.x & 1) x.w[5] |= 0xe0000000u;
        if (threadIdx.x & 2) x.w[5] &= 0x1fffffffu;
        const gf192 g = gf_mul(x, u), v = V < 0 ? g : mulv<(V < 0 ? 0 : V)>(x, u);
        uint64_t bad = 0;
        for (int w = 0; w < 6; ++w) bad |= (g.w[w] ^ v.w[w]);
        out[i] = bad;
        return;
    }
    for (int r = 0; r < CHAIN; ++r) {
        if (V < 0) x = gf_mul(x, u); else x = mulv<(V < 0 ? 0 : V)>(x, u);
    }
    gf_store(out, i, x);
    if (dummy_lds[0] == 0x12345) out[0] = 1;
}

static int lds_for_waves(int w) { return w >= 8 ? 0 : (160 * 1024) / w - ((160 * 1024) / w) % 256 - 1024 * (w == 1 ? 60 : 0); }

template<typename K, typename... Args>
static float time_kernel(K kern, int waves, int lds, Args... args)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int blocks = 256 * waves * 4;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template<int P> static void run_issue(const char *name, uint32_t *d, double valu_per_round)
{
    printf("%-46s", name);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_issue<P>, 256, lds_for_waves(w)));
        const float ms = time_kernel(k_issue<P>, w, lds_for_waves(w), d, 3u);
        const double winstr = (double)256 * w * 4 * 4 * ROUNDS * valu_per_round;      // wave-instructions over the chip
        // cycles per VALU wave-instruction per SIMD at 2.4 GHz: time * clock * SIMDs / instructions
        printf("  w%d(occ %d): %5.2f", w, nb, ms * 1e-3 * 2.4e9 * 1024 / winstr);
    }
    printf("   [cycles @2.4GHz per VALU per SIMD]\n");
}

template<int P> static void run_plain(const char *name, uint32_t *d, double valu_per_round)
{
    printf("%-46s", name);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_plain<P>, 256, lds_for_waves(w)));
        const float ms = time_kernel(k_plain<P>, w, lds_for_waves(w), d, 3u);
        const double winstr = (double)256 * w * 4 * 4 * ROUNDS * valu_per_round;
        printf("  w%d(occ %d): %5.2f", w, nb, ms * 1e-3 * 2.4e9 * 1024 / winstr);
    }
    printf("   [cycles @2.4GHz per VALU per SIMD]\n");
}

template<int V> static void run_mul(const char *name, const uint64_t *in, uint64_t *out, uint64_t *hout, int maxw)
{
    // correctness
    hipLaunchKernelGGL(k_mul<V>, dim3(64), dim3(256), 0, 0, in, out, 1);
    CK(hipMemcpy(hout, out, 64 * 256 * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int i = 0; i < 64 * 256; ++i) bad += hout[i] != 0;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)k_mul<V>));
    printf("%-10s vgprs %3d  %s ", name, fa.numRegs, bad ? "MISMATCH" : "ok");
    for (int w = 1; w <= maxw; ++w) {
        int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mul<V>, 256, lds_for_waves(w)));
        const float ms = time_kernel(k_mul<V>, w, lds_for_waves(w), in, out, 0);
        printf("  w%d(occ %d): %.3e/s", w, nb, (double)256 * w * 4 * 256 * CHAIN / ms * 1e3);
    }
    printf("\n");
}

int main(int argc, char **argv)
{
    const bool products_only = argc > 1 && !strcmp(argv[1], "products");
    uint32_t *d; CK(hipMalloc(&d, (size_t)256 * 8 * 4 * 256 * 4));
    uint64_t *in, *out; CK(hipMalloc(&in, (size_t)65536 * 24)); CK(hipMalloc(&out, (size_t)256 * 8 * 4 * 256 * 24));
    uint64_t *h = (uint64_t *)malloc((size_t)65536 * 24);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < 65536 * 3; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s; }
    CK(hipMemcpy(in, h, (size_t)65536 * 24, hipMemcpyHostToDevice));
    if (!products_only) {
    printf("== issue patterns (cycles at 2.4 GHz per VALU wave-instruction per SIMD; SIMD-32 floor = 2.0) ==\n");
    run_plain<0>("plain v_xor x42", d, 42);
    run_plain<1>("plain alignbit chain x48", d, 48);
    run_issue<0>("rel v_xor x42, no SALU", d, 42);
    run_issue<1>("rel v_xor 6x(7 + set_idx)", d, 42);
    run_issue<2>("rel v_xor 6x(7 + bfe,set_idx)", d, 42);
    run_issue<3>("rel v_xor 6x(7 + bfe,mul,set_idx)", d, 42);
    run_issue<10>("round: 42 xor + shift12, no SALU", d, 54);
    run_issue<11>("round: 6x(7 + set_idx) + shift12", d, 54);
    run_issue<12>("round: 6x(7 + bfe,set_idx) + shift12", d, 54);
    run_issue<13>("round: 6x(7 + bfe,mul,set_idx) + shift12", d, 54);
    }
    printf("== whole products (uniform-multiplier products per second, chip-wide) ==\n");
    run_mul<-1>("general(u unif)", in, out, h, 4);
    run_mul<-2>("general", in, out, h, 4);
    run_mul<0>("v0", in, out, h, 3);
    run_mul<1>("v1", in, out, h, 3);
    run_mul<2>("v2", in, out, h, 3);
    run_mul<5>("v3hi", in, out, h, 3);
    run_mul<3>("v3", in, out, h, 4);
    run_mul<4>("v4", in, out, h, 4);
    run_mul<6>("j0", in, out, h, 8);
    run_mul<7>("j1", in, out, h, 8);
    run_mul<8>("j2", in, out, h, 8);
    return 0;
}
