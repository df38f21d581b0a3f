// Micro-benchmark (MI355X): where the cycles of the EDGE passes go (k_bfly_edge_multi / k_bfly_edge_fwd_batch of libiop_amd/csrc/fft_add.hip: the six
// butterfly levels whose twiddles differ from lane to lane).  The general GF(2^192) product costs 3209 cycles per wave in registers
// (tools/ubench/mul_rates) and about 3900 per wave-butterfly inside the kernel; this rebuilds the kernel's loop with its parts removable:
//   G0  the general levels (pair bits 5..2) as shipped: twiddle = LDS block term ^ LDS shift term, both elements from LDS, general product, two puts,
//       a workgroup barrier per level; 1024-element tile, 256 threads, no register bound (the compiler's choice, as the shipped kernel)
//   G1  G0 with the twiddle held in registers (no twiddle read, no shift-term XOR)
//   G2  G1 with the two elements held in registers (the product and the two additions only)
//   G3  G0 without the workgroup barrier between the levels (wavefront-scope ordering only)
//   G4  G0 bounded to 128 VGPRs (four wavefronts per SIMD)
//   G5  G0 with the 54-register product (gf_mul_lean), bounded to 80 VGPRs (six wavefronts)
//   G6  G0 with both elements of a butterfly fetched as ONE 128-bit + ONE 64-bit LDS read each (array-of-structures tile, 24-byte elements)
//   P0  the whole forward pass: tile from HBM in block order (16 runs of 64 elements), six levels (four general, the two-word and the one-word
//       numerator levels), natural-order (bit-reversed) store
//   P1  P0 with the general product replaced by an XOR (everything of the pass but the four general products)
//   P2  P0 with all six products replaced by XORs (loads, LDS traffic, barriers, stores only)
// Output: cycles per wave-butterfly per SIMD at 2.4 GHz and the clock the run really had.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I libiop_amd/csrc/include -mllvm -pragma-unroll-threshold=1000000 tools/ubench/edge_loop.hip -o tools/ubench/edge_loop
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
// array-of-structures tile: element li at words 3 li .. 3 li + 2
__device__ __forceinline__ gf192 aos_get(const uint64_t *s, int li)
{
    const uint64_t *q = s + 3 * li;
    const uint64_t a = q[0], b = q[1], c = q[2];
    gf192 r;
    r.w[0] = (uint32_t)a; r.w[1] = (uint32_t)(a >> 32); r.w[2] = (uint32_t)b; r.w[3] = (uint32_t)(b >> 32); r.w[4] = (uint32_t)c; r.w[5] = (uint32_t)(c >> 32);
    return r;
}
__device__ __forceinline__ void aos_put(uint64_t *s, int li, const gf192 &v)
{
    uint64_t *q = s + 3 * li;
    q[0] = (uint64_t)v.w[0] | ((uint64_t)v.w[1] << 32);
    q[1] = (uint64_t)v.w[2] | ((uint64_t)v.w[3] << 32);
    q[2] = (uint64_t)v.w[4] | ((uint64_t)v.w[5] << 32);
}
__device__ __forceinline__ uint32_t bitrev_n(uint32_t x, int bits) { return bits ? __brev(x) >> (32 - bits) : 0; }

#define A_LOW 6
#define C_TOP 4
#define TB (A_LOW + C_TOP)
#define E_TILE (1 << TB)
#define T_TOP (1 << C_TOP)
#define REPS 6

enum { G0, G1, G2, G3, G4, G5, G6, P0, P1, P2 };

template<int V> struct Bounds { static const int waves = 1; };
template<> struct Bounds<G4> { static const int waves = 4; };
template<> struct Bounds<G5> { static const int waves = 6; };

template<int V>
__global__ void __launch_bounds__(256, Bounds<V>::waves) k_edge(const uint64_t *table, const uint64_t *tiles_in, uint64_t *tiles_out, uint64_t *out, unsigned long long *clk, int d)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
    const int tid = threadIdx.x, nt = blockDim.x, E = E_TILE, T = T_TOP;
    const int pmin = 2;
    const int NTW = T * ((1 << (A_LOW - pmin)) - 1);               // 240 block terms
    uint64_t *s = smem, *tw = s + 3 * E, *sh = tw + 3 * NTW;
    uint64_t *sm1 = sh + 3 * A_LOW + 2;
    uint32_t *sm0 = (uint32_t *)(sm1 + (E >> 2));
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) { c0 = clock64(); r0 = wall_clock64(); }
    const int midbits = d - TB;
    const size_t mid = blockIdx.x & (((size_t)1 << midbits) - 1), coset = blockIdx.x >> midbits;
    const bool pass = V >= P0;
    for (int e = tid; e < NTW; e += nt) lds_put(tw, NTW, e, gf_load(table, (size_t)e + (mid << 8)));
    for (int e = tid; e < (E >> 1); e += nt) sm0[e] = (uint32_t)(table[e + mid] | 1);
    for (int e = tid; e < (E >> 2); e += nt) sm1[e] = table[e + 7 + mid] | 1;
    if (tid < 3 * A_LOW + 2) sh[tid] = table[tid + 100 + coset];
    if (pass) {
        const uint64_t *src = tiles_in + 3 * (coset << d);
        for (int e = tid; e < E; e += nt) {
            const int top = e >> A_LOW, lo = e & ((1 << A_LOW) - 1);
            lds_put(s, E, e, gf_load(src, ((size_t)top << (d - C_TOP)) | (mid << A_LOW) | (size_t)lo));
        }
    } else {
        for (int li = tid; li < E; li += nt) { gf192 v; for (int k = 0; k < 6; ++k) v.w[k] = (uint32_t)(li * 2654435761u + k * 40503u + blockIdx.x); if (V == G6) aos_put(s, li, v); else lds_put(s, E, li, v); }
    }
    __syncthreads();
    gf192 keep_a = lds_get(s, E, tid), keep_b = lds_get(s, E, tid + 256);
    gf192 tw_fixed = lds_get(tw, NTW, tid & 127);
    for (int rep = 0; rep < (pass ? 1 : REPS); ++rep) {
        for (int t = 0; t < (pass ? A_LOW : A_LOW - pmin); ++t) {
            const int pbit = A_LOW - 1 - t;
            if (pbit == 0) {
                const uint32_t yc = (uint32_t)sh[3 * A_LOW];
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    gf192 a = lds_get(s, E, bf << 1), b = lds_get(s, E, (bf << 1) | 1);
                    if (V == P2) gf_add_to(a, b); else gf_add_to(a, gf_mul_small_over_xk(b, sm0[bf] ^ yc, 24));
                    gf_add_to(b, a);
                    lds_put(s, E, bf << 1, a); lds_put(s, E, (bf << 1) | 1, b);
                }
            } else if (pbit == 1) {
                const uint64_t yc = sh[3 * A_LOW + 1];
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    const int ia = ((bf >> 1) << 2) | (bf & 1);
                    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ia | 2);
                    const uint64_t y = sm1[bf >> 1] ^ yc;
                    if (V == P2) gf_add_to(a, b); else gf_add_to(a, gf_mul_small2_over(b, (uint32_t)y, (uint32_t)(y >> 32), 46, 1));
                    gf_add_to(b, a);
                    lds_put(s, E, ia, a); lds_put(s, E, ia | 2, b);
                }
            } else {
                const uint64_t s0 = sh[3 * pbit], s1 = sh[3 * pbit + 1], s2 = sh[3 * pbit + 2];
                const int tbase = T * ((1 << (A_LOW - 1 - pbit)) - 1);
#pragma unroll 1
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    const int low = bf & ((1 << pbit) - 1), high = bf >> pbit;
                    const int ia = (high << (pbit + 1)) | low, ib = ia | (1 << pbit);
                    gf192 tt;
                    if (V == G1 || V == G2) { tt = tw_fixed; tt.w[0] ^= (uint32_t)bf; }
                    else {
                        tt = lds_get(tw, NTW, tbase + high);
                        tt.w[0] ^= (uint32_t)s0; tt.w[1] ^= (uint32_t)(s0 >> 32); tt.w[2] ^= (uint32_t)s1; tt.w[3] ^= (uint32_t)(s1 >> 32);
                        tt.w[4] ^= (uint32_t)s2; tt.w[5] ^= (uint32_t)(s2 >> 32);
                    }
                    if (V == G2) {
                        gf_add_to(keep_a, gf_mul(keep_b, tt));
                        gf_add_to(keep_b, keep_a);
                    } else if (V == G6) {
                        gf192 a = aos_get(s, ia), b = aos_get(s, ib);
                        gf_add_to(a, gf_mul(b, tt));
                        gf_add_to(b, a);
                        aos_put(s, ia, a); aos_put(s, ib, b);
                    } else {
                        gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ib);
                        if (V == P1 || V == P2) gf_add_to(a, gf_add(b, tt));
                        else if (V == G5) gf_add_to(a, gf_mul_lean(b, tt));
                        else gf_add_to(a, gf_mul(b, tt));
                        gf_add_to(b, a);
                        lds_put(s, E, ia, a); lds_put(s, E, ib, b);
                    }
                }
            }
            if (V == G3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            else if (V != G2) __syncthreads();
        }
    }
    if (V == G2) { lds_put(s, E, tid, keep_a); lds_put(s, E, tid + 256, keep_b); }
    __syncthreads();
    if (pass) {
        uint64_t *dst = tiles_out + 3 * (coset << d);
        for (int sidx = tid; sidx < E; sidx += nt) {
            const int tp = sidx & (T - 1), lo = (sidx >> C_TOP) & ((1 << A_LOW) - 1);
            const int top = (int)bitrev_n((uint32_t)tp, C_TOP);
            const size_t v = ((size_t)bitrev_n((uint32_t)lo, A_LOW) << (d - A_LOW)) | ((size_t)bitrev_n((uint32_t)mid, midbits) << C_TOP) | (size_t)tp;
            gf_store(dst, v, lds_get(s, E, (top << A_LOW) | lo));
        }
    } else {
        gf192 acc = V == G6 ? aos_get(s, tid) : lds_get(s, E, tid);
        gf_add_to(acc, V == G6 ? aos_get(s, tid + 512) : lds_get(s, E, tid + 512));
        gf_store(out, (size_t)blockIdx.x * nt + tid, acc);
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

// The pass as k_bfly_edge_multi runs it: CPW cosets of one tile position per workgroup, one after the other.
//   M0  load -> six levels -> store, coset after coset (as shipped)
//   M1  the next coset's tile is requested into REGISTERS before the current coset's levels and written to LDS after its store: the load latency
//       hides behind the workgroup's own products
//   M2  M1 and the finished tile leaves through registers as well: its LDS reads happen before the next tile's LDS writes, the global stores
//       drain while the next coset computes (no barrier between store and the next load-to-LDS beyond the one that orders the LDS accesses)
enum { M0 = 20, M1, M2 };
#define CPW 4
template<int V>
__global__ void __launch_bounds__(256) k_edge_multi(const uint64_t *table, const uint64_t *tiles_in, uint64_t *tiles_out, unsigned long long *clk, int d, int ncos)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
    const int tid = threadIdx.x, nt = 256, E = E_TILE, T = T_TOP;
    const int pmin = 2;
    const int NTW = T * ((1 << (A_LOW - pmin)) - 1);
    uint64_t *s = smem, *tw = s + 3 * E, *sh = tw + 3 * NTW;
    uint64_t *sm1 = sh + 3 * A_LOW + 2;
    uint32_t *sm0 = (uint32_t *)(sm1 + (E >> 2));
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) { c0 = clock64(); r0 = wall_clock64(); }
    const int midbits = d - TB;
    const size_t mid = blockIdx.x & (((size_t)1 << midbits) - 1), cbase = (size_t)(blockIdx.x >> midbits) * CPW;
    for (int e = tid; e < NTW; e += nt) lds_put(tw, NTW, e, gf_load(table, (size_t)e + (mid << 8)));
    for (int e = tid; e < (E >> 1); e += nt) sm0[e] = (uint32_t)(table[e + mid] | 1);
    for (int e = tid; e < (E >> 2); e += nt) sm1[e] = table[e + 7 + mid] | 1;
    gf192 pre[E_TILE / 256];
    auto request = [&](size_t coset) {
        const uint64_t *src = tiles_in + 3 * (coset << d);
#pragma unroll
        for (int k = 0; k < E_TILE / 256; ++k) {
            const int e = tid + k * 256, top = e >> A_LOW, lo = e & ((1 << A_LOW) - 1);
            pre[k] = gf_load(src, ((size_t)top << (d - C_TOP)) | (mid << A_LOW) | (size_t)lo);
        }
    };
    if (V != M0) request(cbase);
    for (int j = 0; j < CPW; ++j) {
        const size_t coset = cbase + j;
        if (coset >= (size_t)ncos) break;
        if (tid < 3 * A_LOW + 2) sh[tid] = table[tid + 100 + coset];
        if (V == M0) request(coset);
#pragma unroll
        for (int k = 0; k < E_TILE / 256; ++k) lds_put(s, E, tid + k * 256, pre[k]);
        __syncthreads();
        if (V != M0 && j + 1 < CPW && coset + 1 < (size_t)ncos) request(coset + 1);
        for (int t = 0; t < A_LOW; ++t) {
            const int pbit = A_LOW - 1 - t;
            if (pbit == 0) {
                const uint32_t yc = (uint32_t)sh[3 * A_LOW];
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    gf192 a = lds_get(s, E, bf << 1), b = lds_get(s, E, (bf << 1) | 1);
                    gf_add_to(a, gf_mul_small_over_xk(b, sm0[bf] ^ yc, 24));
                    gf_add_to(b, a);
                    lds_put(s, E, bf << 1, a); lds_put(s, E, (bf << 1) | 1, b);
                }
            } else if (pbit == 1) {
                const uint64_t yc = sh[3 * A_LOW + 1];
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    const int ia = ((bf >> 1) << 2) | (bf & 1);
                    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ia | 2);
                    const uint64_t y = sm1[bf >> 1] ^ yc;
                    gf_add_to(a, gf_mul_small2_over(b, (uint32_t)y, (uint32_t)(y >> 32), 46, 1));
                    gf_add_to(b, a);
                    lds_put(s, E, ia, a); lds_put(s, E, ia | 2, b);
                }
            } else {
                const uint64_t s0 = sh[3 * pbit], s1 = sh[3 * pbit + 1], s2 = sh[3 * pbit + 2];
                const int tbase = T * ((1 << (A_LOW - 1 - pbit)) - 1);
#pragma unroll 1
                for (int bf = tid; bf < (E >> 1); bf += nt) {
                    const int low = bf & ((1 << pbit) - 1), high = bf >> pbit;
                    const int ia = (high << (pbit + 1)) | low, ib = ia | (1 << pbit);
                    gf192 tt = lds_get(tw, NTW, tbase + high);
                    tt.w[0] ^= (uint32_t)s0; tt.w[1] ^= (uint32_t)(s0 >> 32); tt.w[2] ^= (uint32_t)s1; tt.w[3] ^= (uint32_t)(s1 >> 32);
                    tt.w[4] ^= (uint32_t)s2; tt.w[5] ^= (uint32_t)(s2 >> 32);
                    gf192 a = lds_get(s, E, ia), b = lds_get(s, E, ib);
                    gf_add_to(a, gf_mul(b, tt));
                    gf_add_to(b, a);
                    lds_put(s, E, ia, a); lds_put(s, E, ib, b);
                }
            }
            __syncthreads();
        }
        uint64_t *dst = tiles_out + 3 * (coset << d);
        for (int sidx = tid; sidx < E; sidx += nt) {
            const int tp = sidx & (T - 1), lo = (sidx >> C_TOP) & ((1 << A_LOW) - 1);
            const int top = (int)bitrev_n((uint32_t)tp, C_TOP);
            const size_t v = ((size_t)bitrev_n((uint32_t)lo, A_LOW) << (d - A_LOW)) | ((size_t)bitrev_n((uint32_t)mid, midbits) << C_TOP) | (size_t)tp;
            gf_store(dst, v, lds_get(s, E, (top << A_LOW) | lo));
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

static uint64_t *g_table, *g_in, *g_out, *g_small;
static unsigned long long *g_clk;

template<int V> void run(const char *name)
{
    const int d = 20, cosets = 32;
    const bool pass = V >= P0;
    const int blocks = pass ? cosets << (d - TB) : 256 * 4 * 6;
    const size_t lds = (size_t)24 * E_TILE + 24 * 240 + 8 * (3 * A_LOW + 2) + 8 * (E_TILE >> 2) + 4 * (E_TILE >> 1);
    CK(hipFuncSetAttribute((const void *)k_edge<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)k_edge<V>));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_edge<V>, dim3(blocks), dim3(256), lds, 0, g_table, g_in, g_out, g_small, g_clk, d);
    CK(hipEventRecord(e0, 0));
    const int launches = pass ? 4 : 1;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_edge<V>, dim3(blocks), dim3(256), lds, 0, g_table, g_in, g_out, g_small, g_clk, d);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= launches;
    const double levels = pass ? A_LOW : (A_LOW - 2) * REPS;
    const double butterflies = (double)blocks * (E_TILE / 2) * levels;
    unsigned long long clk[2];
    CK(hipMemcpy(clk, g_clk, 16, hipMemcpyDeviceToHost));
    const double ghz = clk[1] ? (double)clk[0] / ((double)clk[1] * 10.0) : 0.0;
    printf("%-68s %8.3f ms  %3d VGPRs  %6.0f cycles per wave-butterfly per SIMD at 2.4 GHz%s; workgroup 0 saw %.2f GHz\n", name, ms, fa.numRegs,
           ms * 1e-3 * 2.4e9 * 1024 / (butterflies / 64), pass ? " (mean over the six levels)" : "", ghz);
}

template<int V> void run_multi(const char *name)
{
    const int d = 20, cosets = 32;
    const int blocks = (cosets / CPW) << (d - TB);
    const size_t lds = (size_t)24 * E_TILE + 24 * 240 + 8 * (3 * A_LOW + 2) + 8 * (E_TILE >> 2) + 4 * (E_TILE >> 1);
    CK(hipFuncSetAttribute((const void *)k_edge_multi<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)k_edge_multi<V>));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_edge_multi<V>, dim3(blocks), dim3(256), lds, 0, g_table, g_in, g_out, g_clk, d, cosets);
    CK(hipEventRecord(e0, 0));
    const int launches = 4;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_edge_multi<V>, dim3(blocks), dim3(256), lds, 0, g_table, g_in, g_out, g_clk, d, cosets);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= launches;
    const double butterflies = (double)cosets * ((size_t)1 << (d - 1)) * A_LOW;
    unsigned long long clk[2];
    CK(hipMemcpy(clk, g_clk, 16, hipMemcpyDeviceToHost));
    printf("%-68s %8.3f ms  %3d VGPRs  %6.0f cycles per wave-butterfly per SIMD at 2.4 GHz (mean over the six levels); workgroup 0 saw %.2f GHz\n", name, ms, fa.numRegs,
           ms * 1e-3 * 2.4e9 * 1024 / (butterflies / 64), clk[1] ? (double)clk[0] / ((double)clk[1] * 10.0) : 0.0);
}

int main()
{
    const size_t n = (size_t)1 << 25;
    CK(hipMalloc(&g_table, ((size_t)1 << 20) * 24)); CK(hipMemset(g_table, 0x5a, ((size_t)1 << 20) * 24));
    CK(hipMalloc(&g_in, n * 24)); CK(hipMalloc(&g_out, n * 24)); CK(hipMalloc(&g_small, (size_t)256 * 4 * 6 * 256 * 24)); CK(hipMalloc(&g_clk, 16));
    CK(hipMemset(g_in, 0x17, n * 24));
    for (int pass = 0; pass < 3; ++pass) {
        run<G0>("G0 general levels as shipped");
        run<G1>("G1 twiddle in registers");
        run<G2>("G2 twiddle and elements in registers (product + 2 additions)");
        run<G3>("G3 as shipped, no workgroup barrier");
        run<G4>("G4 as shipped, bounded to 128 VGPRs (4 waves/SIMD)");
        run<G5>("G5 54-register product, bounded to 80 VGPRs (6 waves/SIMD)");
        run<G6>("G6 array-of-structures tile (24-byte elements)");
        run<P0>("P0 whole pass: block-order load, 6 levels, natural-order store");
        run<P1>("P1 whole pass, the four general products replaced by XORs");
        run<P2>("P2 whole pass, all six products replaced by XORs");
        run_multi<M0>("M0 four cosets per workgroup, one after the other (as shipped)");
        run_multi<M1>("M1 next coset's tile requested into registers before the levels");
    }
    return 0;
}
