// Micro-benchmark (MI355X): what separates k_bfly_upper's butterfly loop from the comb product's in-register rate.  The loop of
// libiop_amd/csrc/fft_add.hip (k_bfly_upper<false, true, false>: 2048-element tile in LDS as three 64-bit planes, 512 threads, one butterfly
// per trip, wave-uniform twiddle through the scalar unit, a barrier per level) is rebuilt here with its parts removable:
//   V0  the loop as shipped: twiddle = table[block] ^ shift[level] through s_load (table of 2^19 entries: scalar-cache misses)
//   V1  twiddle from a 64-entry table (scalar-cache hits)
//   V2  twiddle held in SGPRs across the loop (no scalar load at all)
//   V3  V2 and the two elements stay in registers (no LDS traffic, no barrier): the product and the two additions only
//   V4  V0 with no barrier between levels (wavefront-local order only; the result is not a transform, the instruction stream is the same)
//   V6  V0 as a whole pass: every workgroup loads its tile from HBM, runs the five levels ONCE and stores the tile (k_bfly_upper's life)
// The shader clock is read beside the constant 100 MHz counter, so the cycle figures use the frequency the run really had.
// Output: cycles per wave-butterfly per SIMD at 2.4 GHz, 6 waves per SIMD (3 workgroups of 512 threads per CU by LDS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I libiop_amd/csrc/include -mllvm -pragma-unroll-threshold=1000000 tools/ubench/bfly_loop.hip -o tools/ubench/bfly_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../libiop_amd/csrc/gf192_dev.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ gf192 lds_get(const uint64_t *s, int E, int li)
{
    const uint64_t a = s[li], b = s[E + li], c = s[2 * E + li];
    gf192 r;
    r.w[0] = (uint32_t)a; r.w[1] = (uint32_t)(a >> 32); r.w[2] = (uint32_t)b; r.w[3] = (uint32_t)(b >> 32); r.w[4] = (uint32_t)c; r.w[5] = (uint32_t)(c >> 32);
    return r;
}
__device__ __forceinline__ void lds_put(uint64_t *s, int E, int li, const gf192 &v)
{
    s[li] = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32);
    s[E + li] = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32);
    s[2 * E + li] = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
}
__device__ __forceinline__ gf192 tw_load(const uint64_t *table, size_t index, const uint64_t *shift, int level)
{
    const uint64_t *t = table + 3 * index, *r = shift + 3 * level;
    const uint64_t w0 = uniform_load64(t) ^ uniform_load64(r), w1 = uniform_load64(t + 1) ^ uniform_load64(r + 1), w2 = uniform_load64(t + 2) ^ uniform_load64(r + 2);
    gf192 tw;
    tw.w[0] = (uint32_t)w0; tw.w[1] = (uint32_t)(w0 >> 32); tw.w[2] = (uint32_t)w1; tw.w[3] = (uint32_t)(w1 >> 32); tw.w[4] = (uint32_t)w2; tw.w[5] = (uint32_t)(w2 >> 32);
    return tw;
}

#define LEVELS 5
#define REPS 8
template<int V>
__global__ void __launch_bounds__(512, 6) k_loop(const uint64_t *table, size_t table_mask, const uint64_t *shift, uint64_t *out, const uint64_t *tiles_in, uint64_t *tiles_out, unsigned long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
    uint64_t *s = smem;
    const int tid = threadIdx.x, nt = blockDim.x, E = 2048, c = 6;
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) { c0 = clock64(); r0 = wall_clock64(); }
    if (V >= 6) {
        // k_bfly_upper's tile: 32 rows of 64 consecutive elements, the rows 2^14 elements apart
        const size_t base = ((size_t)(blockIdx.x >> 8) << 19) | ((size_t)(blockIdx.x & 255) << 6);
        for (int li = tid; li < E; li += nt) lds_put(s, E, li, gf_load(tiles_in, base | ((size_t)(li >> 6) << 14) | (size_t)(li & 63)));
    } else {
        for (int li = tid; li < E; li += nt) { gf192 v; for (int k = 0; k < 6; ++k) v.w[k] = (uint32_t)(li * 2654435761u + k * 40503u + blockIdx.x); lds_put(s, E, li, v); }
    }
    __syncthreads();
    gf192 keep_a = lds_get(s, E, tid), keep_b = lds_get(s, E, tid + 512);
    gf192 tw_fixed = tw_load(table, 5, shift, 0);
    for (int rep = 0; rep < (V >= 6 ? 1 : REPS); ++rep) {
        for (int t = 0; t < LEVELS; ++t) {
            const int pl = c + LEVELS - 1 - t;                  // local pair bit: 10 .. 6
#pragma unroll 1
            for (int bf = tid; bf < (E >> 1); bf += nt) {
                const int ia = ((bf >> pl) << (pl + 1)) | (bf & ((1 << pl) - 1));
                const uint32_t ia_u = __builtin_amdgcn_readfirstlane((uint32_t)(ia & ~63));
                const size_t block = (((size_t)blockIdx.x << 5) | (ia_u >> (pl + 1))) + ((size_t)1 << (14 + t));
                gf192 tw;
                if (V == 0 || V == 4 || V >= 6) tw = tw_load(table, block & table_mask, shift, t);
                else if (V == 1) tw = tw_load(table, block & 63, shift, t);
                else tw = tw_fixed;
                if (V == 3) {
                    gf_add_to(keep_a, gf_mul_uniform(keep_b, tw));
                    gf_add_to(keep_b, keep_a);
                } else {
                    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ia | (1 << pl));
                    gf_add_to(a, gf_mul_uniform(b, tw));
                    gf_add_to(b, a);
                    lds_put(s, E, ia, a);
                    lds_put(s, E, ia | (1 << pl), b);
                }
            }
            if (V == 4) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            else if (V != 3) __syncthreads();
        }
    }
    if (V == 3) { lds_put(s, E, tid, keep_a); lds_put(s, E, tid + 512, keep_b); }
    __syncthreads();
    if (V >= 6) {
        const size_t base = ((size_t)(blockIdx.x >> 8) << 19) | ((size_t)(blockIdx.x & 255) << 6);
        for (int li = tid; li < E; li += nt) gf_store(tiles_out, base | ((size_t)(li >> 6) << 14) | (size_t)(li & 63), lds_get(s, E, li));
    } else {
        gf192 acc = lds_get(s, E, tid);
        gf_add_to(acc, lds_get(s, E, tid + 1024));
        gf_store(out, (size_t)blockIdx.x * nt + tid, acc);
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

static uint64_t *g_tiles_in, *g_tiles_out;
static unsigned long long *g_clk;
static int g_v6_tiles = 1 << 14;
template<int V> void run(const char *name, const uint64_t *table, size_t mask, const uint64_t *shift, uint64_t *out)
{
    const int blocks = V >= 6 ? g_v6_tiles : 256 * 3 * 8;            // V6: 2^25 elements = 2^14 tiles
    const size_t lds = 48 * 1024;
    CK(hipFuncSetAttribute((const void *)k_loop<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_loop<V>, dim3(blocks), dim3(512), lds, 0, table, mask, shift, out, g_tiles_in, g_tiles_out, g_clk);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_loop<V>, dim3(blocks), dim3(512), lds, 0, table, mask, shift, out, g_tiles_in, g_tiles_out, g_clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double butterflies = (double)blocks * 1024 * LEVELS * (V >= 6 ? 1 : REPS);
    unsigned long long clk[2];
    CK(hipMemcpy(clk, g_clk, 16, hipMemcpyDeviceToHost));
    const double ghz = clk[1] ? (double)clk[0] / ((double)clk[1] * 10.0) : 0.0;          // wall_clock64 ticks at 100 MHz
    printf("%-50s %8.3f ms  %.3e butterflies/s  %6.0f cycles per wave-butterfly per SIMD at 2.4 GHz; workgroup 0 saw %.2f GHz\n", name, ms, butterflies / ms * 1e3,
           ms * 1e-3 * 2.4e9 * 1024 / (butterflies / 64), ghz);
}

int main()
{
    const size_t entries = (size_t)1 << 19;
    uint64_t *table, *shift, *out;
    CK(hipMalloc(&table, entries * 24)); CK(hipMalloc(&shift, 64 * 24)); CK(hipMalloc(&out, (size_t)256 * 3 * 8 * 512 * 24));
    CK(hipMemset(table, 0x5a, entries * 24)); CK(hipMemset(shift, 0x33, 64 * 24));
    CK(hipMalloc(&g_tiles_in, ((size_t)1 << 25) * 24)); CK(hipMalloc(&g_tiles_out, ((size_t)1 << 25) * 24)); CK(hipMalloc(&g_clk, 16));
    CK(hipMemset(g_tiles_in, 0x17, ((size_t)1 << 25) * 24));
    for (int pass = 0; pass < 3; ++pass) {
        for (int tiles : {16384, 5120, 3565, 1024, 768}) {          // launch size: the tail of a launch idles the CUs that finish early
            g_v6_tiles = tiles;
            char label[96];
            snprintf(label, sizeof label, "V6 whole pass, %d tiles in the launch", tiles);
            run<6>(label, table, entries - 1, shift, out);
        }
        g_v6_tiles = 1 << 14;
        run<0>("V0 as shipped (table 2^19 entries)", table, entries - 1, shift, out);
        run<1>("V1 twiddles from 64 entries", table, entries - 1, shift, out);
        run<2>("V2 twiddle in SGPRs", table, entries - 1, shift, out);
        run<3>("V3 + elements in registers (no LDS)", table, entries - 1, shift, out);
        run<4>("V4 as shipped, no workgroup barrier", table, entries - 1, shift, out);
        run<6>("V6 whole pass: HBM load, 5 levels, HBM store", table, entries - 1, shift, out);
    }
    // sustained load: the same whole pass back to back for about four seconds, the rate and the shader clock of every half second — does the
    // chip hold the clock of a short burst under a long one? (the prover's kernels run for tens of milliseconds at a stretch)
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t lds = 48 * 1024;
        for (int window = 0; window < 8; ++window) {
            const int launches = 500;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_loop<6>, dim3(1 << 14), dim3(512), lds, 0, table, entries - 1, shift, out, g_tiles_in, g_tiles_out, g_clk);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long clk[2];
            CK(hipMemcpy(clk, g_clk, 16, hipMemcpyDeviceToHost));
            const double butterflies = (double)launches * (1 << 14) * 1024 * LEVELS;
            printf("sustained V6, window %d: %8.1f ms for %d launches  %.3e butterflies/s  %6.0f cycles per wave-butterfly per SIMD at 2.4 GHz; workgroup 0 saw %.2f GHz\n", window, ms, launches,
                   butterflies / ms * 1e3, ms * 1e-3 * 2.4e9 * 1024 / (butterflies / 64), clk[1] ? (double)clk[0] / ((double)clk[1] * 10.0) : 0.0);
        }
    }
    // launch tails: the same 64 launches of 4096 tiles (k_bfly_upper's typical launch) on ONE stream, and alternating between TWO streams — does the
    // second stream's work fill the CUs that the tail of a launch leaves idle?
    {
        hipStream_t st[2];
        CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
        hipEvent_t e0, e1, j; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
        const size_t lds = 48 * 1024;
        for (int tiles : {4096, 2048, 1024}) {
            for (int streams = 1; streams <= 2; ++streams) {
                for (int pass = 0; pass < 2; ++pass) {
                    const int launches = 64;
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, st[0]));
                    if (streams == 2) { CK(hipEventRecord(j, st[0])); CK(hipStreamWaitEvent(st[1], j, 0)); }
                    for (int i = 0; i < launches; ++i)
                        hipLaunchKernelGGL(k_loop<6>, dim3(tiles), dim3(512), lds, st[streams == 2 ? (i & 1) : 0], table, entries - 1, shift, out, g_tiles_in, g_tiles_out, g_clk);
                    if (streams == 2) { CK(hipEventRecord(j, st[1])); CK(hipStreamWaitEvent(st[0], j, 0)); }
                    CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    const double butterflies = (double)launches * tiles * 1024 * LEVELS;
                    if (pass) printf("%d launches of %5d tiles on %d stream(s): %8.3f ms  %.3e butterflies/s  %6.0f cycles per wave-butterfly per SIMD at 2.4 GHz\n", launches, tiles, streams, ms,
                                     butterflies / ms * 1e3, ms * 1e-3 * 2.4e9 * 1024 / (butterflies / 64));
                }
            }
        }
    }
    return 0;
}