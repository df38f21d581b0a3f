// iopx_comm: what the multi-GPU provers exchange between the ranks (include/libiop_amd.h, "multi-GPU").
//
// Two transports behind one C interface:
//   * RCCL over xGMI.  librccl is resolved at run time with dlopen — the copy that is already in the process when the host program
//     loaded one (PyTorch-ROCm ships its own librccl.so.1; a second copy in one process would be a second set of communicator
//     state), /opt/rocm's otherwise — so the library itself keeps linking against the HIP runtime only.  Collectives take device
//     pointers and are enqueued on the library's stream: the kernels before and after them are ordered by that stream, nothing waits
//     on the host.
//   * callbacks of the host program (its own communicator, or the CPU test-suite's gloo group): pointers are passed through as given.
//
// The reference has no counterpart: libiop's prover is one process (libiop/snark/aurora_snark.tcc:119-146).  The sharding that decides
// WHAT is exchanged is libiop_amd/cpp/dist.hpp.  Host-only code: the two small kernels at the end move rows between layouts.
#include <dlfcn.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "runtime.h"

namespace {

// The part of RCCL's C API used here, restated from <rccl/rccl.h> (NCCL's stable ABI): opaque communicator, 128-byte unique id
// passed by value, enum values of the data types / reductions used.
typedef void *ncclComm_t;
struct ncclUniqueId { char internal[IOPX_COMM_UNIQUE_ID_BYTES]; };
enum { NCCL_UINT8 = 1, NCCL_UINT64 = 5, NCCL_SUM = 0, NCCL_MIN = 3 };

struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllToAll)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};

std::mutex g_api_mu;
RcclApi g_api;
std::atomic_uint_fast64_t g_num_collectives{ 0 }, g_comm_bytes{ 0 };

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return IOPX_OK;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);          // a copy the process already holds wins
        if (h) break;
    }
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return iopx::fail(IOPX_ERR_RUNTIME, "librccl not found (%s): multi-GPU runs need RCCL, or a communicator made with iopx_comm_create_callbacks", dlerror());
    RcclApi a;
    a.handle = h;
#define IOPX_RCCL_SYM(field, sym)                                                                               \
    *(void **)(&a.field) = dlsym(h, sym);                                                                       \
    if (!a.field) return iopx::fail(IOPX_ERR_RUNTIME, "librccl lacks %s", sym)
    IOPX_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    IOPX_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    IOPX_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    IOPX_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    IOPX_RCCL_SYM(AllGather, "ncclAllGather");
    IOPX_RCCL_SYM(AllReduce, "ncclAllReduce");
    IOPX_RCCL_SYM(Broadcast, "ncclBroadcast");
    IOPX_RCCL_SYM(AllToAll, "ncclAllToAll");
    IOPX_RCCL_SYM(Send, "ncclSend");
    IOPX_RCCL_SYM(Recv, "ncclRecv");
    IOPX_RCCL_SYM(GroupStart, "ncclGroupStart");
    IOPX_RCCL_SYM(GroupEnd, "ncclGroupEnd");
#undef IOPX_RCCL_SYM
    g_api = a;
    return IOPX_OK;
}

#define IOPX_RCCL(call)                                                                                           \
    do {                                                                                                          \
        const int r_ = (call);                                                                                    \
        if (r_ != 0) return iopx::fail(IOPX_ERR_RUNTIME, "%s failed: %s", #call, g_api.GetErrorString ? g_api.GetErrorString(r_) : "?"); \
    } while (0)

} // namespace

struct iopx_comm {
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    bool use_callbacks = false;
    iopx_comm_callbacks cb{};
    // rank `rank` of `world` played ALONE (iopx_comm_create_replay): no peer exists; every collective moves, on the library's stream, the bytes this
    // rank would receive — from its own send buffer — so that the stream's work is what the rank's stream would carry between the collectives.
    bool replay = false;
};

namespace iopx {
// per host thread, like the distribution context of libiop_amd/cpp/dist.hpp (dist::ctx()) whose scopes bind and unbind it: a transform on
// another thread never joins a collective this thread's peers did not issue
static thread_local iopx_comm *g_transform_comm = nullptr;
CommInfo transform_comm()
{
    CommInfo ci{ g_transform_comm, 0, 1 };
    if (g_transform_comm) { ci.rank = g_transform_comm->rank; ci.world = g_transform_comm->world; }
    return ci;
}
} // namespace iopx

namespace {

__global__ void k_interleave(const uint64_t *src, size_t parts, size_t count, size_t words, uint64_t *dst)
{
    const size_t total = parts * count * words;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t w = t % words, e = t / words, r = e % parts, i = e / parts;      // destination element e = i * parts + r
        dst[t] = src[(r * count + i) * words + w];
    }
}

__global__ void k_gather_rows(const uint64_t *const *srcs, size_t num_srcs, size_t words, const uint64_t *src_index, const uint64_t *dst_row, size_t count,
                              uint64_t *out)
{
    const size_t total = count * num_srcs * words;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t w = t % words, k = (t / words) % num_srcs, i = t / (words * num_srcs);
        out[(dst_row[i] * num_srcs + k) * words + w] = srcs[k][src_index[i] * words + w];
    }
}

int check_comm(const iopx_comm *c)
{
    if (!c) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null communicator");
    return IOPX_OK;
}

} // namespace

extern "C" {

int iopx_comm_rccl_unique_id(uint8_t *unique_id)
{
    if (!unique_id) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = load_rccl();
    if (rc != IOPX_OK) return rc;
    ncclUniqueId id;
    IOPX_RCCL(g_api.GetUniqueId(&id));
    std::memcpy(unique_id, id.internal, IOPX_COMM_UNIQUE_ID_BYTES);
    return IOPX_OK;
}

int iopx_comm_create_rccl(int rank, int world, const uint8_t *unique_id, iopx_comm **out)
{
    if (!unique_id || !out) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (world < 1 || rank < 0 || rank >= world || (world & (world - 1))) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "rank %d of %d: the world size must be a power of two", rank, world);
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    rc = load_rccl();
    if (rc != IOPX_OK) return rc;
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, IOPX_COMM_UNIQUE_ID_BYTES);
    iopx_comm *c = new iopx_comm();
    c->rank = rank; c->world = world;
    const int r = g_api.CommInitRank(&c->nccl, world, id, rank);
    if (r != 0) { delete c; return iopx::fail(IOPX_ERR_RUNTIME, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_api.GetErrorString(r)); }
    *out = c;
    return IOPX_OK;
}

int iopx_comm_create_callbacks(int rank, int world, const iopx_comm_callbacks *callbacks, iopx_comm **out)
{
    if (!callbacks || !out) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (world < 1 || rank < 0 || rank >= world || (world & (world - 1))) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "rank %d of %d: the world size must be a power of two", rank, world);
    if (world > 1 && (!callbacks->all_gather || !callbacks->all_reduce_u64 || !callbacks->broadcast || !callbacks->all_to_all || !callbacks->sendrecv))
        return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_comm_create_callbacks: every collective must be provided");
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    iopx_comm *c = new iopx_comm();
    c->rank = rank; c->world = world; c->use_callbacks = true; c->cb = *callbacks;
    *out = c;
    return IOPX_OK;
}

int iopx_comm_create_replay(int rank, int world, iopx_comm **out)
{
    if (!out) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (world < 1 || rank < 0 || rank >= world || (world & (world - 1))) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "rank %d of %d: the world size must be a power of two", rank, world);
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    iopx_comm *c = new iopx_comm();
    c->rank = rank; c->world = world; c->replay = true;
    *out = c;
    return IOPX_OK;
}

int iopx_comm_is_replay(const iopx_comm *comm) { return comm && comm->replay ? 1 : 0; }

int iopx_comm_bind_transforms(iopx_comm *comm)
{
    iopx::g_transform_comm = comm;
    return IOPX_OK;
}

int iopx_comm_destroy(iopx_comm *comm)
{
    if (!comm) return IOPX_OK;
    if (iopx::g_transform_comm == comm) iopx::g_transform_comm = nullptr;
    if (comm->nccl) {
        (void)hipStreamSynchronize(iopx::stream());          // collectives in flight hold the communicator
        (void)g_api.CommDestroy(comm->nccl);
    }
    delete comm;
    return IOPX_OK;
}

int iopx_comm_rank(const iopx_comm *comm, int *rank, int *world)
{
    if (!comm) { if (rank) *rank = 0; if (world) *world = 1; return IOPX_OK; }
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return IOPX_OK;
}

int iopx_comm_all_gather_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank)
{
    int rc = check_comm(comm);
    if (rc != IOPX_OK) return rc;
    if (bytes_per_rank == 0) return IOPX_OK;
    if (!d_send || !d_recv) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    ++g_num_collectives; g_comm_bytes += bytes_per_rank;
    if (comm->replay) {                                            // every rank's slot <- this rank's part
        for (int r = 0; r < comm->world; ++r) {
            void *slot = (char *)d_recv + (size_t)r * bytes_per_rank;
            if (slot != d_send) { const int crc_ = iopx::copy_d2d(slot, d_send, bytes_per_rank); if (crc_ != IOPX_OK) return crc_; }
        }
        return IOPX_OK;
    }
    if (comm->use_callbacks) {
        if (comm->world == 1) { if (d_send != d_recv) { const int crc_ = iopx::copy_d2d(d_recv, d_send, bytes_per_rank); if (crc_ != IOPX_OK) return crc_; } return IOPX_OK; }
        rc = comm->cb.all_gather(comm->cb.user, d_send, d_recv, bytes_per_rank, (void *)iopx::stream());
        return rc == 0 ? IOPX_OK : iopx::fail(IOPX_ERR_RUNTIME, "all_gather callback failed (%d)", rc);
    }
    IOPX_RCCL(g_api.AllGather(d_send, d_recv, bytes_per_rank, NCCL_UINT8, comm->nccl, iopx::stream()));
    return IOPX_OK;
}

int iopx_comm_all_reduce_u64_dev(iopx_comm *comm, void *d_buf, size_t count, int op)
{
    int rc = check_comm(comm);
    if (rc != IOPX_OK) return rc;
    if (count == 0) return IOPX_OK;
    if (!d_buf) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (op != IOPX_COMM_SUM && op != IOPX_COMM_MIN) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "unknown reduction");
    ++g_num_collectives; g_comm_bytes += count * 8;
    if (comm->replay) return IOPX_OK;                              // the peers contribute the reduction's neutral element
    if (comm->use_callbacks) {
        if (comm->world == 1) return IOPX_OK;
        rc = comm->cb.all_reduce_u64(comm->cb.user, d_buf, count, op, (void *)iopx::stream());
        return rc == 0 ? IOPX_OK : iopx::fail(IOPX_ERR_RUNTIME, "all_reduce callback failed (%d)", rc);
    }
    IOPX_RCCL(g_api.AllReduce(d_buf, d_buf, count, NCCL_UINT64, op == IOPX_COMM_SUM ? NCCL_SUM : NCCL_MIN, comm->nccl, iopx::stream()));
    return IOPX_OK;
}

int iopx_comm_broadcast_dev(iopx_comm *comm, void *d_buf, size_t bytes, int root)
{
    int rc = check_comm(comm);
    if (rc != IOPX_OK) return rc;
    if (bytes == 0) return IOPX_OK;
    if (!d_buf) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (root < 0 || root >= comm->world) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "broadcast root %d of %d ranks", root, comm->world);
    ++g_num_collectives; if (comm->rank == root) g_comm_bytes += bytes;
    if (comm->replay) return IOPX_OK;                              // a receiver keeps whatever its buffer holds (timing only)
    if (comm->use_callbacks) {
        if (comm->world == 1) return IOPX_OK;
        rc = comm->cb.broadcast(comm->cb.user, d_buf, bytes, root, (void *)iopx::stream());
        return rc == 0 ? IOPX_OK : iopx::fail(IOPX_ERR_RUNTIME, "broadcast callback failed (%d)", rc);
    }
    IOPX_RCCL(g_api.Broadcast(d_buf, d_buf, bytes, NCCL_UINT8, root, comm->nccl, iopx::stream()));
    return IOPX_OK;
}

int iopx_comm_all_to_all_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank)
{
    int rc = check_comm(comm);
    if (rc != IOPX_OK) return rc;
    if (bytes_per_rank == 0) return IOPX_OK;
    if (!d_send || !d_recv || d_send == d_recv) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "all_to_all needs two distinct buffers");
    ++g_num_collectives; g_comm_bytes += bytes_per_rank * (size_t)comm->world;
    if (comm->replay) { { const int crc_ = iopx::copy_d2d(d_recv, d_send, bytes_per_rank * (size_t)comm->world); if (crc_ != IOPX_OK) return crc_; } return IOPX_OK; }
    if (comm->use_callbacks) {
        if (comm->world == 1) { { const int crc_ = iopx::copy_d2d(d_recv, d_send, bytes_per_rank); if (crc_ != IOPX_OK) return crc_; } return IOPX_OK; }
        rc = comm->cb.all_to_all(comm->cb.user, d_send, d_recv, bytes_per_rank, (void *)iopx::stream());
        return rc == 0 ? IOPX_OK : iopx::fail(IOPX_ERR_RUNTIME, "all_to_all callback failed (%d)", rc);
    }
    IOPX_RCCL(g_api.AllToAll(d_send, d_recv, bytes_per_rank, NCCL_UINT8, comm->nccl, iopx::stream()));
    return IOPX_OK;
}

int iopx_comm_sendrecv_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes, int peer)
{
    int rc = check_comm(comm);
    if (rc != IOPX_OK) return rc;
    if (bytes == 0) return IOPX_OK;
    if (!d_send || !d_recv) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null buffer");
    if (peer < 0 || peer >= comm->world || peer == comm->rank) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "peer %d (this is rank %d of %d)", peer, comm->rank, comm->world);
    ++g_num_collectives; g_comm_bytes += bytes;
    if (comm->replay) { if (d_send != d_recv) { const int crc_ = iopx::copy_d2d(d_recv, d_send, bytes); if (crc_ != IOPX_OK) return crc_; } return IOPX_OK; }
    if (comm->use_callbacks) {
        rc = comm->cb.sendrecv(comm->cb.user, d_send, d_recv, bytes, peer, (void *)iopx::stream());
        return rc == 0 ? IOPX_OK : iopx::fail(IOPX_ERR_RUNTIME, "sendrecv callback failed (%d)", rc);
    }
    IOPX_RCCL(g_api.GroupStart());
    const int s = g_api.Send(d_send, bytes, NCCL_UINT8, peer, comm->nccl, iopx::stream());
    const int r = g_api.Recv(d_recv, bytes, NCCL_UINT8, peer, comm->nccl, iopx::stream());
    IOPX_RCCL(g_api.GroupEnd());
    IOPX_RCCL(s);
    IOPX_RCCL(r);
    return IOPX_OK;
}

int iopx_comm_stats(uint64_t *num_collectives, uint64_t *bytes, int reset)
{
    if (num_collectives) *num_collectives = g_num_collectives.load();
    if (bytes) *bytes = g_comm_bytes.load();
    if (reset) { g_num_collectives = 0; g_comm_bytes = 0; }
    return IOPX_OK;
}

int iopx_interleave_dev(const void *d_src, size_t parts, size_t count, size_t elem_bytes, void *d_dst)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    if (parts == 0 || count == 0) return IOPX_OK;
    if (!d_src || !d_dst || d_src == d_dst) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "iopx_interleave_dev needs two distinct buffers");
    if (elem_bytes == 0 || elem_bytes % 8) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "element size must be a multiple of 8 bytes");
    const size_t total = parts * count * (elem_bytes / 8);
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    iopx::ProfScope ps("k_interleave", 2 * total * 8);
    hipLaunchKernelGGL(k_interleave, dim3(blocks), dim3(256), 0, iopx::stream(), (const uint64_t *)d_src, parts, count, elem_bytes / 8, (uint64_t *)d_dst);
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

int iopx_gather_rows_dev(const void *const *d_srcs, size_t num_srcs, size_t elem_bytes, const uint64_t *src_index, const uint64_t *dst_row, size_t count,
                         void *d_out)
{
    int rc = iopx::ensure_device();
    if (rc != IOPX_OK) return rc;
    if (count == 0 || num_srcs == 0) return IOPX_OK;
    if (!d_srcs || !src_index || !dst_row || !d_out) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null argument");
    if (elem_bytes == 0 || elem_bytes % 8) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "element size must be a multiple of 8 bytes");
    // one small device block: the source pointers, then the two index arrays
    std::vector<uint64_t> host(num_srcs + 2 * count);
    for (size_t k = 0; k < num_srcs; ++k) {
        if (!d_srcs[k]) return iopx::fail(IOPX_ERR_INVALID_ARGUMENT, "null source");
        host[k] = (uint64_t)(uintptr_t)d_srcs[k];
    }
    std::memcpy(host.data() + num_srcs, src_index, count * 8);
    std::memcpy(host.data() + num_srcs + count, dst_row, count * 8);
    iopx::TmpBuf meta;
    rc = meta.alloc(host.size() * 8);
    if (rc != IOPX_OK) return rc;
    rc = iopx::upload(meta.p, host.data(), host.size() * 8);
    if (rc != IOPX_OK) return rc;
    const size_t total = count * num_srcs * (elem_bytes / 8);
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 1024);
    iopx::ProfScope ps("k_gather_rows");
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, iopx::stream(), (const uint64_t *const *)meta.u64(), num_srcs, elem_bytes / 8,
                       (const uint64_t *)(meta.u64() + num_srcs), (const uint64_t *)(meta.u64() + num_srcs + count), count, (uint64_t *)d_out);
    IOPX_HIP(hipGetLastError());
    return IOPX_OK;
}

} // extern "C"
