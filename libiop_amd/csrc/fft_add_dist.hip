// ONE additive transform as long as its domain, sharded across N = 2^r GPUs (BASELINE's "codeword sharded across the
// GPUs of one node with RCCL all-to-all over xGMI for the FFT transpose step"): additive_FFT / additive_IFFT (libiop/algebra/fft.tcc:39-204) of
// 2^m coefficients over a 2^m-point affine subspace, every rank holding the contiguous block [rank 2^m / N, (rank + 1) 2^m / N) of the input and
// of the output.  The native form of libiop_amd/dist.py's distributed_fft / distributed_ifft, over an iopx_comm.
//
// Gao-Mateer's recursion (fft.tcc:55-96) splits the polynomial r times into 2^r sub-polynomials interleaved in the coefficient index
// (sub-polynomial s = index mod 2^r).  Rank rho works on s = rev_r(rho):
//   1. transpose: block-distributed coefficients -> "s-cyclic" layout — ONE all-to-all (RCCL over xGMI);
//   2. the top r levels: twist by the rank's slice of the power table and run the Taylor network over the local index bits
//      (iopx_add_taylor_gf192_dev); the r (r + 1) / 2 network operations that touch the index bits which now identify ranks are XORs of whole
//      (or half) shards exchanged between peer ranks;
//   3. a complete LOCAL transform of the rank's sub-polynomial over the depth-r recursed domain (the single-GPU kernels);
//   4. the last r butterfly levels (fft.tcc:102-120, stride >= the shard size): peers exchange shards, iopx_add_combine_gf192_dev.
// The inverse undoes the four steps in reverse order (every network operation is its own inverse in characteristic 2).
// The provers do not need this — their transforms are never as long as the codeword domain — it is the building block for
// polynomials that are.  Host code + three small layout kernels; the arithmetic is the kernels of fft_add.hip.
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "gf192_host.h"
#include "runtime.h"

namespace iopx {
namespace {

inline size_t rev_bits(size_t x, int bits)
{
    size_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// dst[q * cnt + u] = src[u * N + rev(q)]  (pack = 1: block -> chunks by destination)  /  dst[u * N + rev(q)] = src[q * cnt + u]  (pack = 0)
__global__ void k_dfft_transpose(const uint64_t *src, uint64_t *dst, size_t cnt, int r, int pack)
{
    const size_t N = (size_t)1 << r, total = cnt * N * 3;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t w = t % 3, e = t / 3, q = e / cnt, u = e % cnt;
        size_t s = 0, x = q;
        for (int i = 0; i < r; ++i) { s = (s << 1) | (x & 1); x >>= 1; }
        const size_t a = (q * cnt + u) * 3 + w, b = (u * N + s) * 3 + w;
        if (pack) dst[a] = src[b]; else dst[b] = src[a];
    }
}

// dst[i] = src[2 i + parity]  (the even / odd slots of a shard as a contiguous half)
__global__ void k_dfft_take_half(const uint64_t *src, uint64_t *dst, size_t half, int parity)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < half * 3; t += (size_t)gridDim.x * blockDim.x)
        dst[t] = src[(2 * (t / 3) + (size_t)parity) * 3 + t % 3];
}

// dst[stride * i + offset] ^= src[i]
__global__ void k_dfft_xor(uint64_t *dst, const uint64_t *src, size_t count, size_t stride, size_t offset)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < count * 3; t += (size_t)gridDim.x * blockDim.x)
        dst[(stride * (t / 3) + offset) * 3 + t % 3] ^= src[t];
}

unsigned grid_of(size_t work) { const size_t g = (work + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

struct DistPlan {
    int m = 0, r = 0;
    size_t rank = 0, world = 1, s = 0, n_loc = 0;
    std::vector<std::vector<hgf192>> rec;       // per top level j: the recursed basis of that level (m - 1 - j vectors)
    std::vector<hgf192> rs;                     // per top level j: the recursed shift
    std::vector<hgf192> local_basis;
    hgf192 local_shift;
    std::vector<std::unique_ptr<DevBuf>> twist, twist_inv;      // per top level: this rank's slice of the (inverse) twist powers
};

std::mutex g_dplan_mu;
std::map<std::vector<uint64_t>, std::shared_ptr<DistPlan>> g_dplans;      // shared: a caller keeps its plan alive across an eviction
size_t g_dplan_bytes = 0;                                                  // device bytes of the cached plans' twist tables
const size_t DPLAN_CACHE_BYTES = (size_t)4 << 30;

hgf192 hpow(hgf192 base, size_t e)
{
    hgf192 r = hgf192::one();
    for (; e; e >>= 1) { if (e & 1) r = r * base; base = base.squared(); }
    return r;
}

int get_dist_plan(const uint64_t *basis, int m, const uint64_t *shift, size_t rank, size_t world, std::shared_ptr<DistPlan> *out)
{
    std::vector<uint64_t> key(basis, basis + 3 * (size_t)m);
    key.insert(key.end(), shift, shift + 3);
    key.push_back((uint64_t)m); key.push_back(rank); key.push_back(world);
    std::lock_guard<std::mutex> lk(g_dplan_mu);
    auto it = g_dplans.find(key);
    if (it != g_dplans.end()) { *out = it->second; return IOPX_OK; }
    std::shared_ptr<DistPlan> pl(new DistPlan());
    int r = 0;
    while (((size_t)1 << r) < world) ++r;
    pl->m = m; pl->r = r; pl->rank = rank; pl->world = world; pl->s = rev_bits(rank, r); pl->n_loc = (size_t)1 << (m - r);
    std::vector<hgf192> b;
    for (int i = 0; i < m; ++i) b.push_back(hgf192::from_words(basis + 3 * i));
    hgf192 sh = hgf192::from_words(shift);
    for (int j = 0; j < r; ++j) {                                          // fft.tcc:57-96 for the top r levels
        const hgf192 beta = b[m - 1 - j];
        if (beta.is_zero()) return fail(IOPX_ERR_INVALID_ARGUMENT, "additive FFT: basis vectors are linearly dependent");
        const hgf192 binv = beta.inverse();
        // this rank's slice of the level-j twist: beta^((l 2^r + s) >> j) = beta^(s >> j) (beta^(2^(r-j)))^l
        hgf192 base = beta;
        for (int k = 0; k < r - j; ++k) base = base.squared();
        const hgf192 init = hpow(beta, pl->s >> j);
        for (int inverse = 0; inverse < 2; ++inverse) {
            std::unique_ptr<DevBuf> tab(new DevBuf());
            int rc = tab->alloc(pl->n_loc * 24);
            if (rc != IOPX_OK) return rc;
            const hgf192 bb = inverse ? base.inverse() : base, ii = inverse ? init.inverse() : init;
            rc = iopx_gf192_pow_table_dev(tab->u64(), pl->n_loc, bb.w, ii.w);
            if (rc != IOPX_OK) return rc;
            (inverse ? pl->twist_inv : pl->twist).push_back(std::move(tab));
        }
        std::vector<hgf192> newb;
        for (int i = 0; i < m - 1 - j; ++i) {
            const hgf192 nb = b[i] * binv;
            newb.push_back(nb);
            b[i] = nb.squared() + nb;
        }
        pl->rec.push_back(newb);
        const hgf192 ns = sh * binv;
        pl->rs.push_back(ns);
        sh = ns.squared() + ns;
        b.resize(m - 1 - j);
    }
    pl->local_basis = b;
    pl->local_shift = sh;
    const size_t plan_bytes = 2 * (size_t)r * pl->n_loc * 24;
    if (g_dplans.size() >= 16 || g_dplan_bytes + plan_bytes > DPLAN_CACHE_BYTES) { g_dplans.clear(); g_dplan_bytes = 0; }
    g_dplan_bytes += plan_bytes;
    *out = pl;
    g_dplans[key] = std::move(pl);
    return IOPX_OK;
}

struct Ctx {
    iopx_comm *comm;
    DistPlan *pl;
    int exchange(const void *send, void *recv, size_t bytes, size_t peer_s) const          // peer given by its sub-polynomial index
    {
        return iopx_comm_sendrecv_dev(comm, send, recv, bytes, (int)rev_bits(peer_s, pl->r));
    }
};

int xor_into(uint64_t *dst, const uint64_t *src, size_t count, size_t stride = 1, size_t offset = 0)
{
    hipLaunchKernelGGL(k_dfft_xor, dim3(grid_of(count * 3)), dim3(256), 0, stream(), dst, src, count, stride, offset);
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int take_half(const uint64_t *src, uint64_t *dst, size_t half, int parity)
{
    hipLaunchKernelGGL(k_dfft_take_half, dim3(grid_of(half * 3)), dim3(256), 0, stream(), src, dst, half, parity);
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

// The network operations of one top level on the index bits that identify ranks (bits k + 1, k for k = r - 1 .. j; forward order, or
// k = j .. r - 1 with each operation's two steps swapped for the inverse).  Forward op on quarters (bit k+1, bit k): (1,0) += (1,1); (0,1) += (1,0).
int rank_bit_network(const Ctx &c, uint64_t *S, uint64_t *R, uint64_t *H, int j, bool inverse)
{
    const DistPlan &pl = *c.pl;
    const int r = pl.r;
    const size_t s = pl.s, n_loc = pl.n_loc, half = n_loc / 2;
    int rc;
    for (int step = 0; step < r - j; ++step) {
        const int k = inverse ? j + step : r - 1 - step;
        if (k + 1 < r) {                                    // both bits select ranks: whole-shard XORs between peers
            const int hi = (int)((s >> (k + 1)) & 1), lo = (int)((s >> k) & 1);
            for (int phase = 0; phase < 2; ++phase) {
                const bool first = inverse ? phase == 1 : phase == 0;      // first = the (1,0) += (1,1) half of the operation
                if (first) {
                    if (hi == 1) {
                        if ((rc = c.exchange(S, R, n_loc * 24, s ^ ((size_t)1 << k))) != IOPX_OK) return rc;
                        if (lo == 0 && (rc = xor_into(S, R, n_loc)) != IOPX_OK) return rc;
                    }
                } else if (hi != lo) {                      // (1,0) and (0,1): (0,1) += (1,0)
                    if ((rc = c.exchange(S, R, n_loc * 24, s ^ ((size_t)3 << k))) != IOPX_OK) return rc;
                    if (hi == 0 && (rc = xor_into(S, R, n_loc)) != IOPX_OK) return rc;
                }
            }
        } else {                                            // bit k selects the rank, bit k + 1 is local index bit 0 (odd / even slots)
            const int bit = (int)((s >> k) & 1);
            const size_t peer = s ^ ((size_t)1 << k);
            for (int phase = 0; phase < 2; ++phase) {
                const bool first = inverse ? phase == 1 : phase == 0;
                // first: the holder of (.,0) adds the peer's (1,1) = its odd slots into its own odd slots (1,0);
                // second: the holder of (.,1) adds the peer's (1,0) = its odd slots into its own even slots (0,1)
                if ((rc = take_half(S, H, half, 1)) != IOPX_OK) return rc;
                if ((rc = c.exchange(H, R, half * 24, peer)) != IOPX_OK) return rc;
                if (first && bit == 0 && (rc = xor_into(S, R, half, 2, 1)) != IOPX_OK) return rc;
                if (!first && bit == 1 && (rc = xor_into(S, R, half, 2, 0)) != IOPX_OK) return rc;
            }
        }
    }
    return IOPX_OK;
}

int dist_transform(iopx_comm *comm, const uint64_t *d_block, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *d_out, bool inverse)
{
    int rc = ensure_device();
    if (rc != IOPX_OK) return rc;
    if (!comm || !d_block || !d_out || !basis || !shift) return fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    int rank = 0, world = 1;
    iopx_comm_rank(comm, &rank, &world);
    int r = 0;
    while ((1 << r) < world) ++r;
    // the transpose moves 2^(m - 2r) elements between every pair of ranks: a shorter transform has nothing to put in a chunk
    if (m > 40 || (world > 1 && m < 2 * (size_t)r)) return fail(IOPX_ERR_INVALID_ARGUMENT, "a %zu-dimensional domain cannot be split over %d ranks (needs m >= %d)", m, world, 2 * r);
    if (world == 1) return inverse ? iopx_add_ifft_gf192_dev(d_block, basis, m, shift, d_out) : iopx_add_fft_gf192_dev(d_block, (size_t)1 << m, basis, m, shift, d_out);
    std::shared_ptr<DistPlan> held;
    if ((rc = get_dist_plan(basis, (int)m, shift, (size_t)rank, (size_t)world, &held)) != IOPX_OK) return rc;
    DistPlan *pl = held.get();
    const Ctx c{ comm, pl };
    const size_t n_loc = pl->n_loc, cnt = n_loc / (size_t)world, bytes = n_loc * 24;
    const int lm = (int)m - r;
    TmpBuf bufS, bufR, bufH, bufX;
    if ((rc = bufS.alloc(bytes)) != IOPX_OK || (rc = bufR.alloc(bytes)) != IOPX_OK || (rc = bufH.alloc(bytes / 2 + 24)) != IOPX_OK || (rc = bufX.alloc(bytes)) != IOPX_OK) return rc;
    uint64_t *S = bufS.u64(), *R = bufR.u64(), *H = bufH.u64(), *X = bufX.u64();
    std::vector<uint64_t> lb;
    for (const hgf192 &v : pl->local_basis) lb.insert(lb.end(), v.w, v.w + 3);
    auto level_consts = [&](int lvl, std::vector<uint64_t> &B) {
        B.clear();
        for (const hgf192 &v : pl->rec[lvl]) B.insert(B.end(), v.w, v.w + 3);
    };
    std::vector<uint64_t> B;
    if (!inverse) {
        // 1. transpose to the s-cyclic layout: coefficient rank n_loc + t goes to the rank of sub-polynomial t mod N, local slot rank n_loc / N + t / N
        hipLaunchKernelGGL(k_dfft_transpose, dim3(grid_of(n_loc * 3)), dim3(256), 0, stream(), d_block, X, cnt, r, 1);
        IOPX_HIP(hipGetLastError());
        if ((rc = iopx_comm_all_to_all_dev(comm, X, S, cnt * 24)) != IOPX_OK) return rc;
        // 2. top r levels
        for (int j = 0; j < r; ++j) {
            if ((rc = iopx_add_taylor_gf192_dev(S, (size_t)lm, pl->twist[j]->u64())) != IOPX_OK) return rc;
            if ((rc = rank_bit_network(c, S, R, H, j, false)) != IOPX_OK) return rc;
        }
        // 3. the local transform over the recursed domain
        if ((rc = iopx_add_fft_gf192_dev(S, n_loc, lb.data(), (size_t)lm, pl->local_shift.w, X)) != IOPX_OK) return rc;
        // 4. the last r butterfly levels across blocks
        uint64_t *cur = X, *nxt = S;
        for (int t = 0; t < r; ++t) {
            const int peer = rank ^ (1 << t), upper = (rank >> t) & 1, lvl = r - 1 - t;
            if ((rc = iopx_comm_sendrecv_dev(comm, cur, R, bytes, peer)) != IOPX_OK) return rc;
            level_consts(lvl, B);
            uint64_t *out = (t == r - 1) ? d_out : nxt;
            rc = iopx_add_combine_gf192_dev(upper ? R : cur, upper ? cur : R, out, n_loc, ((size_t)rank & (((size_t)1 << t) - 1)) * n_loc, B.data(), pl->rec[lvl].size(),
                                            pl->rs[lvl].w, upper);
            if (rc != IOPX_OK) return rc;
            nxt = cur; cur = out;
        }
        return IOPX_OK;
    }
    // inverse: 4'. the last r butterfly levels across blocks, outermost first
    const uint64_t *cur = d_block;
    uint64_t *spare[2] = { X, S };
    for (int t = r - 1; t >= 0; --t) {
        const int peer = rank ^ (1 << t), upper = (rank >> t) & 1, lvl = r - 1 - t;
        if ((rc = iopx_comm_sendrecv_dev(comm, cur, R, bytes, peer)) != IOPX_OK) return rc;
        level_consts(lvl, B);
        uint64_t *out = spare[t & 1];
        rc = iopx_add_combine_inv_gf192_dev(upper ? R : cur, upper ? cur : R, out, n_loc, ((size_t)rank & (((size_t)1 << t) - 1)) * n_loc, B.data(), pl->rec[lvl].size(),
                                            pl->rs[lvl].w, upper);
        if (rc != IOPX_OK) return rc;
        cur = out;
    }
    // 3'. the local inverse transform (cur is X or S; the result goes to the other one)
    uint64_t *loc = (cur == X) ? S : X;
    if ((rc = iopx_add_ifft_gf192_dev(cur, lb.data(), (size_t)lm, pl->local_shift.w, loc)) != IOPX_OK) return rc;
    // 2'. top r levels, innermost first
    for (int j = r - 1; j >= 0; --j) {
        if ((rc = rank_bit_network(c, loc, R, H, j, true)) != IOPX_OK) return rc;
        if ((rc = iopx_add_taylor_inv_gf192_dev(loc, (size_t)lm, pl->twist_inv[j]->u64())) != IOPX_OK) return rc;
    }
    // 1'. transpose back: chunk q of the s-cyclic shard belongs to rank q
    uint64_t *back = (loc == X) ? S : X;
    if ((rc = iopx_comm_all_to_all_dev(comm, loc, back, cnt * 24)) != IOPX_OK) return rc;
    hipLaunchKernelGGL(k_dfft_transpose, dim3(grid_of(n_loc * 3)), dim3(256), 0, stream(), (const uint64_t *)back, d_out, cnt, r, 0);
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // namespace

void clear_dist_plans()
{
    std::lock_guard<std::mutex> lk(g_dplan_mu);
    g_dplans.clear();
    g_dplan_bytes = 0;
}
} // namespace iopx

extern "C" {

int iopx_add_fft_gf192_dist_dev(iopx_comm *comm, const uint64_t *d_block_coeffs, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *d_block_out)
{
    return iopx::dist_transform(comm, d_block_coeffs, basis, m, shift, d_block_out, false);
}

int iopx_add_ifft_gf192_dist_dev(iopx_comm *comm, const uint64_t *d_block_evals, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *d_block_out)
{
    return iopx::dist_transform(comm, d_block_evals, basis, m, shift, d_block_out, true);
}

} // extern "C"
