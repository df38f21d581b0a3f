// The provers over N GPUs, one process per GPU (BASELINE configs[3] and configs[4]).
//
// The reference prover is one process with every oracle in one std::vector (libiop/snark/aurora_snark.tcc:119-146).  Here every rank runs
// the SAME prover code (aurora.hpp / fractal.hpp) and holds 1/N of every vector over the codeword domain L and over the FRI domains L^(i):
//
//   affine subspaces (GF(2^192))          rank g holds positions [g |D| / N, (g + 1) |D| / N): the affine sub-domain spanned by the first
//                                         m - log2 N basis vectors, shifted by element_by_index(g 2^(m - log2 N)) (subspace.tcc:56-71).  A
//                                         low-degree extension is the rank's coset range of the transform, FRI cosets and Merkle leaves are
//                                         contiguous runs (subspace.tcc:73-91) and stay on one rank.
//   multiplicative cosets (181-bit field)  rank r holds the positions p = r (mod N) at local index p / N: the sub-coset (shift g^r) <g^N>, a
//                                         multiplicative coset in its own right.  An FRI coset / Merkle leaf {j + k n / 2^eta}
//                                         (subgroup.tcc:175-197) lies on rank j mod N whole; leaf digests are exchanged once per tree
//                                         (all-to-all) so that each rank builds a contiguous sub-tree.
//
// A domain carries the mark (field_subset::distributed()); the device operators of aurora.hpp / fractal.hpp act on dist::local_domain(D)
// and the BCS layer (iop.hpp) assembles roots, authentication paths and query answers across the ranks.  What is exchanged per proof:
// N x 32-byte sub-roots per tree (all-gather), the sumcheck polynomial's coefficients (one broadcast), FRI tails below 64 elements per rank
// (all-gather), answers and paths (one all-reduce per tree each: every row of a zeroed buffer has exactly one owner), the proof-of-work
// hit (min all-reduce).  Everything over the small domains (<= 2^20 elements), the hashchain and the query bookkeeping are replicated: every
// rank derives the same challenges and returns the same transcript.
//
// dist::scope binds a communicator (iopx_comm, include/libiop_amd.h) for the calls of this host thread; without one the same code is the
// single-GPU prover (rank 0 of 1, nothing marked).
#pragma once
#include <cstdlib>
#include <algorithm>
#include <functional>
#include "device.hpp"

namespace libiop_amd {
namespace dist {

struct context {
    iopx_comm *comm = nullptr;
    std::size_t rank = 0, world = 1, log_world = 0;
    bool active() const { return comm != nullptr; }      // a one-rank communicator still takes the distributed code path
};
inline context &ctx() { static thread_local context c; return c; }

class scope {
    context saved_;
public:
    explicit scope(iopx_comm *comm) : saved_(ctx())
    {
        context c;
        if (comm) {
            int r = 0, w = 1;
            check(iopx_comm_rank(comm, &r, &w));
            c.comm = comm; c.rank = (std::size_t)r; c.world = (std::size_t)w; c.log_world = detail::log2_ceil((std::size_t)w);
        }
        ctx() = c;
        check(iopx_comm_bind_transforms(c.comm));            // the replicated transforms split their phase 1 over the ranks (include/libiop_amd.h)
    }
    scope(const scope &) = delete;
    scope &operator=(const scope &) = delete;
    ~scope() { ctx() = saved_; (void)iopx_comm_bind_transforms(saved_.comm); }
};

// A stretch of work that only some ranks execute (rank 0 interpolating for everybody): transforms inside it must not take part in collectives.
class one_rank_section {
public:
    one_rank_section() { check(iopx_comm_bind_transforms(nullptr)); }
    one_rank_section(const one_rank_section &) = delete;
    one_rank_section &operator=(const one_rank_section &) = delete;
    ~one_rank_section() { (void)iopx_comm_bind_transforms(ctx().comm); }
};

static const std::size_t MIN_BLOCK = 64;      // elements per rank below which a domain is kept whole on every rank

// Can vectors over D, committed in Merkle leaves of coset_size elements, be split over the ranks?  Subspaces: enough elements per rank.
// Cosets: also at least N leaves per rank, so that the digest exchange splits evenly.
template<typename FieldT>
bool can_distribute(const field_subset<FieldT> &D, std::size_t coset_size)
{
    const context &c = ctx();
    if (!c.active()) return false;
    const std::size_t n = D.num_elements(), W = c.world;
    if (n / W < MIN_BLOCK || coset_size == 0 || n < coset_size) return false;
    if (D.type() == affine_subspace_type) return (n / W) % coset_size == 0;
    return (n / coset_size) % (W * W) == 0;
}

// the codeword domain of a proof; first_coset_size: the Merkle leaves of the first round (2^localization[0]).  When the domain is too
// small for the communicator it stays whole and every rank computes the whole proof.
template<typename FieldT>
field_subset<FieldT> mark_codeword_domain(field_subset<FieldT> D, std::size_t first_coset_size)
{
    D.set_distributed(can_distribute(D, first_coset_size));
    return D;
}

// FRI's domain chain: L^(i+1) stays distributed while its own tree can be; the last domain carries no oracle (the final polynomial is
// interpolated from it on every rank), so the fold into it gathers.
template<typename FieldT>
void mark_fri_domains(std::vector<field_subset<FieldT>> &domains, const std::vector<std::size_t> &localization)
{
    for (std::size_t i = 1; i < domains.size(); ++i) {
        const std::size_t cs = i < localization.size() ? (std::size_t)1 << localization[i] : 1;
        domains[i].set_distributed(domains[i - 1].distributed() && i < localization.size() && can_distribute(domains[i], cs));
    }
}

template<typename FieldT>
std::size_t local_size(const field_subset<FieldT> &D) { return D.distributed() ? D.num_elements() / ctx().world : D.num_elements(); }

// this rank's part of D as a domain of its own (D itself when it is not distributed); `rank` = another rank's part
template<typename FieldT>
field_subset<FieldT> local_domain(const field_subset<FieldT> &D, std::size_t rank = (std::size_t)-1)
{
    if (!D.distributed()) return D;
    const context &c = ctx();
    if (rank == (std::size_t)-1) rank = c.rank;
    if (D.type() == affine_subspace_type) {
        const std::size_t m = D.dimension(), r = c.log_world;
        uint64_t s[3];
        const FieldT D_shift = D.shift();
        std::memcpy(s, detail::words(&D_shift), 24);
        for (std::size_t k = 0; k < r; ++k)
            if ((rank >> k) & 1) for (int w = 0; w < 3; ++w) s[w] ^= detail::words(&D.basis()[m - r + k])[w];
        return field_subset<FieldT>(affine_subspace<FieldT>(std::vector<FieldT>(D.basis().begin(), D.basis().begin() + (m - r)), field_host<FieldT>::from_words(s)));
    }
    // (shift g^rank) <g^N>: the default generator of order n / N is g^N (subgroup.tcc:55-59: multiplicative_generator^((p - 1) / order))
    const FieldT shift = field_host<FieldT>::mul(D.shift(), field_host<FieldT>::pow(D.generator(), rank));
    return field_subset<FieldT>(D.num_elements() / c.world, shift);
}

// the cosets of span(basis[0..d)) of an affine codeword domain this rank holds: (first, count)
template<typename FieldT>
std::pair<std::size_t, std::size_t> coset_range(const field_subset<FieldT> &L, std::size_t d)
{
    const std::size_t cosets = (std::size_t)1 << (L.dimension() - d);
    if (!L.distributed()) return { 0, cosets };
    const context &c = ctx();
    if (cosets % c.world) throw std::invalid_argument("fewer cosets than ranks: this transform needs the exchange steps");
    return { c.rank * (cosets / c.world), cosets / c.world };
}

// ---- windows of a domain: `count` positions first + i * stride that form a domain of their own.  Subspaces: stride 1, `first` a multiple
// of `count` (a power of two) — the coset of the span of the first log2(count) basis vectors through element `first`.  Cosets:
// count * stride = |D|, first < stride — the coset of order `count` through element `first`.  The HEAD of a domain is the window at 0 whose
// evaluations determine a polynomial of `count` coefficients: the positions IFFT_of_known_degree reads (fft.tcc:435-475). ----
struct window {
    std::size_t first, stride, count;
    bool operator==(const window &o) const { return first == o.first && stride == o.stride && count == o.count; }
};

template<typename FieldT>
window head_window(const field_subset<FieldT> &D, std::size_t count)
{
    if (count == 0 || count > D.num_elements() || D.num_elements() % count) throw std::invalid_argument("head_window: the count does not divide the domain");
    return { 0, D.type() == affine_subspace_type ? 1 : D.num_elements() / count, count };
}

// a window of `count` <= head_count positions disjoint from the head of head_count: positions [head_count, head_count + count) of a subspace; the
// coset of order `count` through element stride_head / 2 of a coset (the head holds the multiples of stride_head)
template<typename FieldT>
window beside_head_window(const field_subset<FieldT> &D, std::size_t head_count, std::size_t count)
{
    const window h = head_window(D, head_count);
    if (2 * head_count > D.num_elements() || count > head_count || head_count % count) throw std::invalid_argument("beside_head_window: no room next to the head");
    return D.type() == affine_subspace_type ? window{ head_count, 1, count } : window{ h.stride / 2, D.num_elements() / count, count };
}

// the window as a domain (never distributed)
template<typename FieldT>
field_subset<FieldT> window_domain(const field_subset<FieldT> &D, const window &w)
{
    if (D.type() == affine_subspace_type) {
        const std::size_t d = detail::log2_ceil(w.count);
        if (w.stride != 1 || ((std::size_t)1 << d) != w.count || w.first % w.count || w.first + w.count > D.num_elements()) throw std::invalid_argument("not a window of this subspace");
        uint64_t s[3];                                        // element_by_index(first) (subspace.tcc:56-71)
        const FieldT D_shift = D.shift();
        std::memcpy(s, detail::words(&D_shift), 24);
        for (std::size_t k = d; k < D.dimension(); ++k)
            if ((w.first >> k) & 1) for (int i = 0; i < 3; ++i) s[i] ^= detail::words(&D.basis()[k])[i];
        return field_subset<FieldT>(affine_subspace<FieldT>(std::vector<FieldT>(D.basis().begin(), D.basis().begin() + d), field_host<FieldT>::from_words(s)));
    }
    if (w.count * w.stride != D.num_elements() || w.first >= w.stride) throw std::invalid_argument("not a window of this coset");
    return field_subset<FieldT>(w.count, field_host<FieldT>::mul(D.shift(), field_host<FieldT>::pow(D.generator(), w.first)));      // shift g^first
}
template<typename FieldT>
field_subset<FieldT> head_domain(const field_subset<FieldT> &D, std::size_t count) { return window_domain(D, head_window(D, count)); }

// the rank that holds every position of the window, or (std::size_t)-1 when they are spread over several (any rank when D is whole everywhere)
template<typename FieldT>
std::size_t window_owner(const field_subset<FieldT> &D, const window &w)
{
    if (!D.distributed()) return ctx().rank;
    const std::size_t W = ctx().world;
    if (D.type() == affine_subspace_type) {
        const std::size_t per = D.num_elements() / W;
        return w.first / per == (w.first + w.count - 1) / per ? w.first / per : (std::size_t)-1;
    }
    return w.stride % W == 0 ? w.first % W : (std::size_t)-1;
}
template<typename FieldT>
bool head_on_rank0(const field_subset<FieldT> &D, std::size_t count) { return !D.distributed() || window_owner(D, head_window(D, count)) == 0; }

// v: the owner's part of a vector over D (the whole vector when D is not distributed) -> the vector over the window
template<typename FieldT>
device_vector<FieldT> window_of(const device_vector<FieldT> &v, const field_subset<FieldT> &D, const window &w)
{
    if (window_owner(D, w) != ctx().rank) throw std::logic_error("window_of: this rank does not hold the window");
    const std::size_t W = D.distributed() ? ctx().world : 1;
    if (D.type() == affine_subspace_type) {
        const std::size_t off = w.first % (D.num_elements() / W);
        return off == 0 && w.count == v.size() ? v : v.slice(off, w.count);
    }
    const std::size_t off = w.first / W, stride = w.stride / W;
    if (stride == 1) return v;
    device_vector<FieldT> out(w.count);
    check(iopx_gather_stride_dev(v.slice(off, v.size() - off).data(), w.count, stride, sizeof(FieldT), out.data()));
    return out;
}

// `have`: a vector over the window `held` of D (not distributed: the holder has all of it) -> the vector over the window `want`, when every
// position of `want` is one of `held` (false otherwise)
template<typename FieldT>
bool window_from_window(const device_vector<FieldT> &have, const window &held, const window &want, field_subset_type type, device_vector<FieldT> &out)
{
    if (type == affine_subspace_type) {
        if (want.first < held.first || want.first + want.count > held.first + held.count) return false;
        out = want == held ? have : have.slice(want.first - held.first, want.count);
        return true;
    }
    if (want.first < held.first || (want.first - held.first) % held.stride || want.stride % held.stride || want.count > held.count) return false;
    if (want == held) { out = have; return true; }
    const std::size_t off = (want.first - held.first) / held.stride;
    out = device_vector<FieldT>(want.count);
    check(iopx_gather_stride_dev(have.slice(off, have.size() - off).data(), want.count, want.stride / held.stride, sizeof(FieldT), out.data()));
    return true;
}

// ---- windows produced by the transform that makes the codeword.  Over a multiplicative coset a window is a strided set of positions; gathering it
// from the finished 2^25-point codeword reads a whole 128-byte line for every 24 useful bytes (3.2 x the useful traffic, profiles/r05_traffic_fractal.json).
// The prover that will ask for windows of its codeword-domain oracles says so at registration (bcs_prover::want_window); the forward transform
// (dev::FFT, aurora.hpp) then writes the wanted windows from its last pass (iopx_mul_fft_fp3_windows_dev) and leaves them here, keyed by the codeword
// it made; bcs_prover::submit_oracle adopts them into the oracle's window cache, a round end drops what nobody adopted. ----
template<typename FieldT>
struct window_collector {
    std::size_t domain_elements = 0;                  // the (whole, multiplicative) domain the wanted windows are windows of
    std::vector<window> wanted;                       // at most two (a transform's last pass carries two extra outputs)
    struct entry { device_vector<FieldT> codeword; std::vector<std::pair<window, device_vector<FieldT>>> windows; };
    std::vector<entry> produced;
};
template<typename FieldT>
inline window_collector<FieldT> *&active_collector() { static thread_local window_collector<FieldT> *c = nullptr; return c; }

// ---- collectives on device vectors (enqueued on the library's stream) ------------------------------------------------------------
template<typename T>
device_array<T> all_gather(const device_array<T> &local)                                     // rank-major concatenation
{
    const context &c = ctx();
    device_array<T> out(local.size() * c.world);
    check(iopx_comm_all_gather_dev(c.comm, local.data(), out.data(), local.size() * sizeof(T)));
    return out;
}

// the whole vector, in natural order, from the ranks' parts: rank-major blocks for subspaces, residue classes for cosets
template<typename FieldT>
device_vector<FieldT> gather_layout(const device_vector<FieldT> &local, field_subset_type type)
{
    const context &c = ctx();
    const device_vector<FieldT> parts(all_gather<FieldT>(local));
    if (type == affine_subspace_type || c.world == 1) return parts;
    device_vector<FieldT> out(parts.size());
    check(iopx_interleave_dev(parts.data(), c.world, local.size(), sizeof(FieldT), out.data()));
    return out;
}

template<typename FieldT>
device_vector<FieldT> gather(const device_vector<FieldT> &local, const field_subset<FieldT> &D)
{
    return D.distributed() ? gather_layout<FieldT>(local, D.type()) : local;
}

template<typename FieldT>
void broadcast(const device_vector<FieldT> &v, int root) { check(iopx_comm_broadcast_dev(ctx().comm, v.data(), v.size() * sizeof(FieldT), root)); }

// ---- merkle_tree::get_set_membership_proof's node walk (merkle_tree.tcc:256-336): heap indices of the auxiliary hashes in the
// reference's order — level by level from the leaves, a left node whose right sibling is not queried takes the sibling, a right node
// takes its left sibling ----
inline std::vector<std::size_t> membership_proof_node_indices(std::size_t num_leaves, const std::vector<std::size_t> &positions)
{
    std::vector<std::size_t> out, S = positions;
    std::sort(S.begin(), S.end());
    S.erase(std::unique(S.begin(), S.end()), S.end());
    if (S.empty()) return out;
    for (std::size_t &p : S) {
        if (p >= num_leaves) throw std::invalid_argument("All positions must be between 0 and num_leaves-1.");
        p += num_leaves - 1;
    }
    while (!(S.size() == 1 && S[0] == 0)) {
        std::vector<std::size_t> next;
        for (std::size_t i = 0; i < S.size();) {
            const std::size_t pos = S[i];
            next.push_back((pos - 1) / 2);
            if (pos % 2 == 0) { out.push_back(pos - 1); ++i; }
            else if (i + 1 == S.size() || S[i + 1] != pos + 1) { out.push_back(pos + 1); ++i; }
            else i += 2;
        }
        S.swap(next);
    }
    return out;
}

// pow::solve_pow (bcs/pow.tcc:67-103) split by candidate range: in super-batch s rank r searches candidates [(s N + r) B_s, (s N + r + 1) B_s);
// a min all-reduce of the hits ends the search at the first super-batch that has one, and the minimum is the reference's first hit (the
// candidates of earlier super-batches all failed).  32-byte challenge -> 32-byte answer.
// `behind_the_grind`: host work (and launches) that does not depend on the answer; on one GPU it runs between the enqueueing of the first long batch
// and the wait for it, so the GPU grinds while the host prepares — and afterwards in every other case.
inline std::string solve_pow(const std::string &challenge, std::size_t pow_bitlen, const std::function<void()> &behind_the_grind = nullptr)
{
    const context &c = ctx();
    uint8_t answer[32];
    const uint8_t *ch = reinterpret_cast<const uint8_t *>(challenge.data());
    if (!c.active() || c.world == 1) {
        // iopx_pow_solve_blake2b's batches (2^16 candidates, then 16 x as many each time): the short ones are waited for as before
        uint64_t found = ~(uint64_t)0, first = 0, batch = (uint64_t)1 << 16;
        bool ran = false;
        // batches of at least 2^IOPX_POW_BEHIND_LOG2 candidates take the work behind them (tests: 0)
        const uint64_t long_batch = (uint64_t)1 << std::min(40, std::max(0, iopx_get_option("IOPX_POW_BEHIND_LOG2", 22)));
        while (found == ~(uint64_t)0) {
            if (behind_the_grind && !ran && batch >= long_batch) {
                check(iopx_pow_search_blake2b_begin(ch, pow_bitlen, first, batch));
                struct finish { bool armed = true; ~finish() { uint64_t ignored; if (armed) (void)iopx_pow_search_blake2b_end(&ignored); } } guard;
                behind_the_grind();
                ran = true;
                guard.armed = false;
                check(iopx_pow_search_blake2b_end(&found));
            } else {
                check(iopx_pow_search_blake2b(ch, pow_bitlen, first, batch, &found));
            }
            first += batch;
            if (batch < ((uint64_t)1 << 28)) batch <<= 4;
        }
        if (behind_the_grind && !ran) behind_the_grind();
        check(iopx_pow_candidate_blake2b(ch, found, answer));
        return std::string(reinterpret_cast<const char *>(answer), 32);
    }
    if (behind_the_grind) behind_the_grind();
    const uint64_t none = ~(uint64_t)0;
    uint64_t first = 0, batch = (uint64_t)1 << 14;
    const device_array<uint64_t> d_hit(1);
    const bool replay = iopx_comm_is_replay(c.comm) != 0;            // nobody to hear a hit from: this rank grinds the whole super-batch
    for (;;) {
        uint64_t hit = none;
        check(iopx_pow_search_blake2b(reinterpret_cast<const uint8_t *>(challenge.data()), pow_bitlen, replay ? first : first + c.rank * batch,
                                      replay ? c.world * batch : batch, &hit));
        check(iopx_upload_small(d_hit.data(), &hit, 8));
        check(iopx_comm_all_reduce_u64_dev(c.comm, d_hit.data(), 1, IOPX_COMM_MIN));
        check(iopx_memcpy_d2h(&hit, d_hit.data(), 8));
        if (hit != none) {
            check(iopx_pow_candidate_blake2b(reinterpret_cast<const uint8_t *>(challenge.data()), hit, answer));
            return std::string(reinterpret_cast<const char *>(answer), 32);
        }
        first += c.world * batch;
        if (batch < ((uint64_t)1 << 22)) batch <<= 2;
    }
}

} // namespace dist
} // namespace libiop_amd
