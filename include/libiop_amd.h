/*
 * libiop_amd — C ABI of the MI355X (gfx950) prover hot path for libiop-style IOP SNARKs.
 *
 * The reference (scipr-lab/libiop) has no FFI boundary: its seams are C++ templates.  Each entry point
 * below replaces one of those templates for FieldT = libff::gf192 and is what a thin header shim in
 * libiop would bind.  Paths are relative to the reference tree.
 *
 * Conventions
 *   - A field element is 3 little-endian uint64 words (24 bytes): libff::gf192's in-memory layout.
 *   - `basis` (m elements) and `shift` (1 element) always live in HOST memory: they are O(m) metadata
 *     (libiop/algebra/field_subset/subspace.tcc:47-108).  Evaluation order is the reference's:
 *     output index i is the point shift + sum_{bit k of i} basis[k] (libiop/algebra/utils.tcc:8-30).
 *   - `*_dev` entry points take DEVICE pointers for codeword-sized buffers and enqueue work on the
 *     library's current stream (iopx_set_stream); they do not synchronise.  The variants without the
 *     suffix take HOST pointers, copy in/out and synchronise: drop-in for std::vector callers.
 *   - Every function returns IOPX_OK or a negative code; iopx_last_error() gives the message.  The C++
 *     shim rethrows: INVALID_ARGUMENT -> std::invalid_argument, LOGIC -> std::logic_error,
 *     RUNTIME / NO_DEVICE -> std::runtime_error (the exception types used at fft.tcc:333,368,
 *     merkle_tree.tcc:27-31,98-108, blake2b.tcc:153-156).
 *   - There is no CPU fallback: without a usable HIP device every compute entry point fails with
 *     IOPX_ERR_NO_DEVICE.
 */
#ifndef LIBIOP_AMD_H
#define LIBIOP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IOPX_OK                     0
#define IOPX_ERR_INVALID_ARGUMENT  (-1)
#define IOPX_ERR_LOGIC             (-2)
#define IOPX_ERR_RUNTIME           (-3)
#define IOPX_ERR_NO_DEVICE         (-4)

#define IOPX_DOMAIN_ADDITIVE        0   /* affine_subspace_type   (field_subset.hpp) */
#define IOPX_DOMAIN_MULTIPLICATIVE  1   /* multiplicative_coset_type */

/* ---- runtime ------------------------------------------------------------------------------------ */
int         iopx_version(void);
const char *iopx_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails). */
int         iopx_device_count(void);
/* Bind the calling process to a device (one process per GPU).  The binding is latched by the first call that touches the
 * device: a later iopx_init with another device fails with IOPX_ERR_LOGIC.  Every entry point re-selects the bound device for
 * the calling host thread, but the library keeps per-process state (plans, temporaries): call it from one thread at a time. */
int         iopx_init(int device);
/* Enqueue all subsequent work on the caller's hipStream_t, taken as given — NULL is the HIP legacy default stream (what
 * torch.cuda.current_stream().cuda_stream is for torch's default stream), so library kernels and the caller's own work on that
 * stream are ordered.  iopx_use_own_stream() goes back to the library's private non-blocking stream (the initial state): then
 * the caller orders its producers / consumers against iopx_synchronize() itself. */
int         iopx_set_stream(void *hip_stream);
int         iopx_use_own_stream(void);
int         iopx_synchronize(void);
int         iopx_malloc(void **dptr, size_t bytes);
int         iopx_free(void *dptr);
int         iopx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int         iopx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
/* A second stream for work nothing later in the caller's sequence waits for (the provers put each round's Merkle tree there: its root is
 * needed by the transcript only — reference quirk F8, bcs/hashing/blake2b.tcc:51-66 — while the next round's transforms go on on the main
 * stream).  _begin: the side stream waits for everything enqueued so far and becomes the library's current stream (kernels, copies, pool
 * allocations); _end: back to the main stream, the side work still in flight; _join: the main stream waits for the side stream (no host
 * wait) and buffers freed inside the section become reusable.  iopx_synchronize and iopx_set_stream join first.  Sections do not nest; one
 * host thread per process drives them.  Collectives of a communicator stay on the main stream (the provers do not fork when distributed). */
int         iopx_side_stream_begin(void);
int         iopx_side_stream_end(void);
int         iopx_side_stream_join(void);
/* iopx_memcpy_d2h that takes part in an iopx_defer_downloads window (below): queued there, delivered by iopx_defer_downloads_end */
int         iopx_memcpy_d2h_deferrable(void *dst_host, const void *src_dev, size_t bytes);
/* Transcript extraction reads back two small results per Merkle tree (iopx_query_responses_dev, iopx_merkle_membership_proof_dev).  Between
 * _begin and _end those calls only queue their read-backs; _end drains the stream once and fills every host buffer handed to them in
 * between (the buffers must stay alive until then).  Everything else behaves as usual inside the window. */
int         iopx_defer_downloads_begin(void);
int         iopx_defer_downloads_end(void);
/* Drop every cached per-domain plan (twist-power tables, twiddle tables). */
int         iopx_clear_plans(void);

/* ---- additive FFT / IFFT over GF(2^192) --------------------------------------------------------- */
/* additive_FFT(poly_coeffs, domain): libiop/algebra/fft.tcc:39-124 (dispatch :414-419).
 * Evaluates the polynomial with n_coeffs <= 2^m coefficients (zero padded, fft.tcc:43-44) on the
 * affine subspace (basis[m], shift); writes 2^m evaluations. */
int iopx_add_fft_gf192_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                           const uint64_t *shift, uint64_t *d_out);
int iopx_add_fft_gf192(const uint64_t *coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                       const uint64_t *shift, uint64_t *out);
/* One shard of the same transform (multi-GPU low-degree extension): with
 * d = ceil(log2 n_coeffs) the output splits into 2^(m-d) contiguous blocks of 2^d evaluations (the cosets of
 * span(basis[0..d))); this computes blocks [coset_begin, coset_begin + coset_count) into d_out
 * (coset_count * 2^d elements), i.e. elements [coset_begin * 2^d, ...) of iopx_add_fft_gf192_dev's output. */
int iopx_add_lde_gf192_dev(const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                           const uint64_t *shift, size_t coset_begin, size_t coset_count, uint64_t *d_out);
/* additive_IFFT(evals, domain): libiop/algebra/fft.tcc:126-204 (dispatch :428-433).  2^m in, 2^m out.
 * IFFT_of_known_degree_over_field_subset (fft.tcc:458-475) is this call on the first
 * 2^ceil(log2 degree) evaluations with the first ceil(log2 degree) basis vectors. */
int iopx_add_ifft_gf192_dev(const uint64_t *d_evals, const uint64_t *basis, size_t m, const uint64_t *shift,
                            uint64_t *d_out);
int iopx_add_ifft_gf192(const uint64_t *evals, const uint64_t *basis, size_t m, const uint64_t *shift,
                        uint64_t *out);
/* Batched forms for several vectors over the SAME domain (Aurora transforms f_w, f_Az, f_Bz, f_Cz, ... one after the other,
 * r1cs_rs_iop.tcc:459-568): a 2^20-point transform is latency-bound per pass, so the vectors share every launch.
 *   iopx_add_ifft_gf192_batch_dev   `batch` inverse transforms, vectors back to back (batch * 2^m elements in and out)
 *   iopx_add_lde_gf192_batch_dev    `batch` low-degree extensions: host arrays of device pointers, polynomial k with n_coeffs
 *                                   coefficients -> the coset range of its transform at d_outs[k] (as iopx_add_lde_gf192_dev) */
int iopx_add_ifft_gf192_batch_dev(const uint64_t *d_evals, size_t batch, const uint64_t *basis, size_t m, const uint64_t *shift,
                                  uint64_t *d_out);
int iopx_add_lde_gf192_batch_dev(const uint64_t *const *d_coeffs, size_t n_coeffs, size_t batch, const uint64_t *basis, size_t m,
                                 const uint64_t *shift, size_t coset_begin, size_t coset_count, uint64_t *const *d_outs);

/* FFT_over_field_subset(IFFT_over_field_subset(evals, H), L) for `batch` vectors stored back to back (2^d_dim elements each),
 * H = span(basis[0..d_dim)) + eval_shift, L = span(basis[0..m)) + shift: the composite the reference uses to move evaluations over
 * a systematic domain onto the codeword domain (r1cs_rs_iop.tcc:459-478, basic_lincheck_aux.tcc:94-118, fractal_indexer.tcc:123-156).
 * Writes cosets [coset_begin, +coset_count) of vector k's codeword to d_outs[k] (as iopx_add_lde_gf192_dev).  The coefficient form
 * is never materialised: the shift-independent halves of the two transforms cancel exactly, the outputs are the same field elements. */
int iopx_add_reextend_gf192_batch_dev(const uint64_t *d_evals, size_t batch, const uint64_t *basis, size_t m, size_t d_dim, const uint64_t *eval_shift,
                                      const uint64_t *shift, size_t coset_begin, size_t coset_count, uint64_t *const *d_outs);
/* ... and, in the same call, the codewords of `n_polys` polynomials given by n_coeffs <= 2^d_dim coefficients each (host array of device
 * pointers): the transform of FFT_over_field_subset(coeffs, L) joins the batch after its own coefficient phase, so the last butterfly
 * pass — shared by up to four vectors — serves both kinds (r1cs_rs_iop.tcc:567-568 next to :459-478: f_w with f_Az, f_Bz, f_Cz).
 * d_outs: the `batch` re-extensions first, then the n_polys codewords.  Outputs equal the separate calls' bit for bit. */
int iopx_add_reextend_lde_gf192_batch_dev(const uint64_t *d_evals, size_t batch, const uint64_t *const *d_coeffs, size_t n_coeffs, size_t n_polys,
                                          const uint64_t *basis, size_t m, size_t d_dim, const uint64_t *eval_shift, const uint64_t *shift,
                                          size_t coset_begin, size_t coset_count, uint64_t *const *d_outs);

/* Two groups of evaluation vectors over cosets of span(basis[0..d_dim)) with different shifts, re-extended in one batch (group a's codewords first
 * in d_outs, then group b's): what iopx_add_reextend_gf192_batch_dev does per group, with the forward passes — and the last pass up to four vectors share —
 * run over all of them.  The Aurora prover's round 1: f_Az, f_Bz, f_Cz over H together with f_w over the first coset of V inside L. */
int iopx_add_reextend2_gf192_batch_dev(const uint64_t *d_evals_a, size_t batch_a, const uint64_t *eval_shift_a, const uint64_t *d_evals_b, size_t batch_b,
                                       const uint64_t *eval_shift_b, const uint64_t *basis, size_t m, size_t d_dim, const uint64_t *shift,
                                       size_t coset_begin, size_t coset_count, uint64_t *const *d_outs);

/* Building blocks of ONE transform sharded across GPUs (libiop_amd/dist.py).  The top log2(N) levels of
 * additive_FFT touch index bits that live on different GPUs; dist.py runs them with these calls plus peer exchanges.
 *   iopx_add_taylor_gf192_dev   in place: S[i] *= d_twist[i] (optional), then the Taylor-expansion network of one
 *                               level over all index bits of the shard (fft.tcc:62-83 with j = 0)
 *   iopx_gf192_pow_table_dev    d_out[l] = init * base^l (the shard's slice of a twist-power table)
 *   iopx_add_combine_gf192_dev  one butterfly level across shards (fft.tcc:116-117): tw_i = shift_term + sum_k
 *                               bit_k(index_base + i) * basis[k]; out = a + tw*b (upper = 0) or a + tw*b + b (upper = 1) */
int iopx_add_taylor_gf192_dev(uint64_t *d_S, size_t log_n, const uint64_t *d_twist);
/* ... and their inverses (additive_IFFT, fft.tcc:126-204), for ONE inverse transform sharded across GPUs:
 *   iopx_add_taylor_inv_gf192_dev    in place: the network undone over all index bits of the shard, then S[i] *= d_twist[i] (inverse powers)
 *   iopx_add_combine_inv_gf192_dev   one butterfly level across shards undone: from lo = a + tw*b and up = lo + b, out = a (upper = 0) or b (upper = 1) */
int iopx_add_taylor_inv_gf192_dev(uint64_t *d_S, size_t log_n, const uint64_t *d_twist);
int iopx_add_combine_inv_gf192_dev(const uint64_t *d_lo, const uint64_t *d_up, uint64_t *d_out, size_t count, size_t index_base,
                                   const uint64_t *basis, size_t nb, const uint64_t *shift_term, int upper);
int iopx_gf192_pow_table_dev(uint64_t *d_out, size_t count, const uint64_t *base, const uint64_t *init);
int iopx_add_combine_gf192_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count, size_t index_base,
                               const uint64_t *basis, size_t nb, const uint64_t *shift_term, int upper);

/* ---- multiplicative-coset FFT / IFFT / FRI fold over the 181-bit prime field (libff edwards_Fr) --- */
/* Elements are libff Fp_model Montgomery words (3 x uint64, R = 2^192), exactly the bytes libiop hashes.
 * `gen` is the generator of the order-2^log_n subgroup (multiplicative_coset::generator(), subgroup.tcc:55-59),
 * `shift` the coset shift; evaluation order is natural: index i <-> shift * gen^i.
 *   iopx_mul_fft_fp3      multiplicative_FFT_degree_aware, libiop/algebra/fft.tcc:236-317 (n_coeffs <= 2^log_n,
 *                         any length; only ceil(log2 n_coeffs) butterfly levels run)
 *   iopx_mul_ifft_fp3     multiplicative_IFFT_internal, fft.tcc:343-361 (libfqfft iFFT / icosetFFT)
 *   iopx_mul_ifft_known_degree_fp3_dev   IFFT_of_known_degree_over_field_subset, fft.tcc:435-456: every
 *                         (2^log_n / 2^ceil(log2 degree))-th evaluation, IFFT over that sub-coset; writes
 *                         2^ceil(log2 degree) coefficients
 *   iopx_fri_fold_mul_fp3 multiplicative_evaluate_next_f_i_over_entire_domain, fri_aux.tcc:106-249
 *                         (cosets {j + k * n / coset_size}, subgroup.tcc:175-197) */
int iopx_mul_fft_fp3_dev(const uint64_t *d_coeffs, size_t n_coeffs, size_t log_n, const uint64_t *gen,
                         const uint64_t *shift, uint64_t *d_out);
/* ... and, from the same last pass, up to two WINDOWS of the output: d_windows[w][k] = d_out[window_first[w] + (k << window_log_stride[w])] for
 * k < 2^(log_n - window_log_stride[w]) (window_first[w] < 2^window_log_stride[w]) — the coset of order 2^(log_n - log_stride) through element `first`,
 * e.g. the positions IFFT_of_known_degree_over_field_subset reads (fft.tcc:435-456).  Saves the strided sweep over the finished codeword. */
int iopx_mul_fft_fp3_windows_dev(const uint64_t *d_coeffs, size_t n_coeffs, size_t log_n, const uint64_t *gen, const uint64_t *shift, uint64_t *d_out,
                                 size_t num_windows, const size_t *window_first, const size_t *window_log_stride, uint64_t *const *d_windows);
int iopx_mul_fft_fp3(const uint64_t *coeffs, size_t n_coeffs, size_t log_n, const uint64_t *gen,
                     const uint64_t *shift, uint64_t *out);
int iopx_mul_ifft_fp3_dev(const uint64_t *d_evals, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                          uint64_t *d_out);
int iopx_mul_ifft_fp3(const uint64_t *evals, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                      uint64_t *out);
int iopx_mul_ifft_known_degree_fp3_dev(const uint64_t *d_evals, size_t degree, size_t log_n, const uint64_t *gen,
                                       const uint64_t *shift, uint64_t *d_out);
int iopx_fri_fold_mul_fp3_dev(const uint64_t *d_f_i, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                              size_t coset_size, const uint64_t *x_i, uint64_t *d_next);
int iopx_fri_fold_mul_fp3(const uint64_t *f_i, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                          size_t coset_size, const uint64_t *x_i, uint64_t *next);

/* Host-side scalars of the 181-bit prime field for the domain metadata of the template boundary (no device needed):
 *   iopx_fp3_subgroup_generator        multiplicative_generator^((p - 1) / 2^log_order), the generator of the order-2^log_order
 *                                      subgroup (multiplicative_subgroup_base::construct_internal, subgroup.tcc:55-59)
 *   iopx_fp3_multiplicative_generator  FieldT::multiplicative_generator (element_outside_of_subset, subgroup.tcc:311-315)
 *   iopx_fp3_host_mul / _host_pow      a * b, a^exponent (shift^(2^eta) of the FRI domain chain, fri_ldt.tcc:292-308) */
int iopx_fp3_subgroup_generator(size_t log_order, uint64_t *gen);
int iopx_fp3_multiplicative_generator(uint64_t *gen);
int iopx_fp3_host_mul(const uint64_t *a, const uint64_t *b, uint64_t *out);
int iopx_fp3_host_pow(const uint64_t *a, uint64_t exponent, uint64_t *out);

/* ---- FRI fold over GF(2^192) -------------------------------------------------------------------- */
/* evaluate_next_f_i_over_entire_domain for affine subspaces:
 * libiop/protocols/ldt/fri/fri_aux.tcc:5-34 -> :36-103.  f_i has 2^m evaluations over (basis, shift);
 * coset_size = 2^eta (contiguous cosets, subspace.tcc:73-91); writes 2^m / coset_size values. */
int iopx_fri_fold_add_gf192_dev(const uint64_t *d_f_i, const uint64_t *basis, size_t m, const uint64_t *shift,
                                size_t coset_size, const uint64_t *x_i, uint64_t *d_next);
int iopx_fri_fold_add_gf192(const uint64_t *f_i, const uint64_t *basis, size_t m, const uint64_t *shift,
                            size_t coset_size, const uint64_t *x_i, uint64_t *next);

/* FRI_protocol::compute_domains for affine subspaces (libiop/protocols/ldt/fri/fri_ldt.tcc:310-338): the chain of derived
 * domains L^(1), L^(2), ... — basis q(basis[eta_i..]), shift q(shift) with q the subspace polynomial of the first eta_i basis
 * vectors.  Host-only metadata; out_bases holds the derived bases back to back (sum of their dimensions), out_shifts one
 * element per derived domain. */
int iopx_fri_domains_gf192(const uint64_t *basis, size_t m, const uint64_t *shift, const size_t *localization, size_t num_reductions,
                           uint64_t *out_bases, uint64_t *out_shifts);

/* ---- BCS Merkle tree, BLAKE2b ------------------------------------------------------------------- */
/* merkle_tree::construct_with_leaves_serialized_by_cosets + compute_inner_nodes:
 * libiop/bcs/merkle_tree.tcc:92-151, 200-229 with blake2b_leafhash / blake2b_two_to_one_hash
 * (libiop/bcs/hashing/blake2b.tcc:120-160, blake2b.cpp:28-48), 32-byte digests.
 *   oracles      host array of num_oracles pointers, each to n elements of elem_bytes raw bytes
 *   domain_type  position map of the default domain of size n (IOPX_DOMAIN_*)
 *   salts        NULL, or num_leaves * salt_bytes zk salts: leaf = H(H(slice) || salt)
 *   nodes        (2 L - 1) * 32 bytes, heap order, L = n / coset_size leaves at index (L - 1) + i;
 *                nodes[0..32) is the root (merkle_tree::get_root, merkle_tree.tcc:231-240). */
int iopx_merkle_blake2b_dev(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                            size_t coset_size, int domain_type, const uint8_t *d_salts, size_t salt_bytes,
                            uint8_t *d_nodes);
int iopx_merkle_blake2b(const void *const *oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                        size_t coset_size, int domain_type, const uint8_t *salts, size_t salt_bytes,
                        uint8_t *nodes);
/* The two halves of the construction, for trees whose leaves are hashed on several GPUs (libiop_amd/dist.py):
 *   iopx_merkle_leaves_blake2b_dev   merkle_tree.tcc:116-149 only: the leaf digests, written to nodes[L-1 .. 2L-2]
 *   iopx_merkle_inner_blake2b_dev    compute_inner_nodes (merkle_tree.tcc:200-229) over leaf digests already in place */
int iopx_merkle_leaves_blake2b_dev(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n,
                                   size_t coset_size, int domain_type, const uint8_t *d_salts, size_t salt_bytes,
                                   uint8_t *d_nodes);
int iopx_merkle_inner_blake2b_dev(uint8_t *d_nodes, size_t num_leaves);

/* ---- BCS Merkle tree, Poseidon over alt_bn128 Fr ------------------------------------------------- */
/* A poseidon_params<FieldT> instance (libiop/bcs/hashing/poseidon.hpp:20-60; the shipped sets are
 * poseidon.tcc:311-520, selected by hash_enum.tcc:73-110).  ark / mds hold CANONICAL integers, 4 little-endian
 * 64-bit words each: ark[(full_rounds + partial_rounds) * state_size] round-major, mds[state_size^2] row-major
 * (ignored when near_mds != 0: the add-only mixing layers of poseidon.tcc:195-224).  capacity = state_size - rate = 1. */
typedef struct iopx_poseidon_params {
    size_t alpha;            /* 3, 5 or 17 */
    size_t full_rounds, partial_rounds;
    size_t rate, state_size; /* state_size <= 4 */
    int near_mds;
    const uint64_t *ark;
    const uint64_t *mds;
} iopx_poseidon_params;

/* get_poseidon_parameters (libiop/bcs/hashing/hash_enum.tcc:12-24): the shipped parameter set of a bcs_hash_type
 * (2 = starkware_poseidon_type, 3 = high_alpha_poseidon_type; state_size 0 / 3, or 4 for the high-alpha set, poseidon.hpp:56).
 * The ark / mds pointers refer to static tables inside the library. */
int iopx_poseidon_shipped_params(int bcs_hash_type, size_t state_size, iopx_poseidon_params *out);
/* FieldT(bigint): canonical 4-word integers -> Montgomery words (libff Fp_model::mont_repr), on the device. */
int iopx_bn128_to_montgomery_dev(const uint64_t *d_canonical, uint64_t *d_out, size_t count);
/* poseidon::apply_permutation (poseidon.tcc:273-297) on `count` states of state_size Montgomery elements, in place. */
int iopx_poseidon_permute_bn128_dev(const iopx_poseidon_params *params, uint64_t *d_states, size_t count);
/* merkle_tree<FieldT, FieldT> with algebraic_leafhash / algebraic_two_to_one_hash over a Poseidon sponge
 * (algebraic_sponge.tcc:18-100,220-265; merkle_tree.tcc:92-151,200-229).  Oracles hold Montgomery elements (32 bytes);
 *   salts   NULL, or num_leaves * 32 bytes: the zk salt of leaf i is parsed as algebraic_sponge.tcc:110-125 does and
 *           absorbed after the slice (zk_hash, :232-245)
 *   nodes   (2 L - 1) Montgomery elements, heap order as for iopx_merkle_blake2b. */
int iopx_merkle_poseidon_bn128_dev(const iopx_poseidon_params *params, const void *const *d_oracles, size_t num_oracles,
                                   size_t n, size_t coset_size, int domain_type, const uint8_t *d_salts,
                                   uint64_t *d_nodes);
int iopx_merkle_poseidon_bn128(const iopx_poseidon_params *params, const void *const *oracles, size_t num_oracles,
                               size_t n, size_t coset_size, int domain_type, const uint8_t *salts, uint64_t *nodes);

/* ---- transcript extraction ------------------------------------------------------------------------ */
/* merkle_tree::get_set_membership_proof (libiop/bcs/merkle_tree.tcc:242-336) from a device-resident node array (32-byte
 * digests in heap order: BLAKE2b or Poseidon trees): writes the auxiliary hashes, in the reference's order, to host memory.
 * positions: leaf indices in any order, duplicates allowed (:256-258).  *num_aux receives the count; fails with
 * IOPX_ERR_INVALID_ARGUMENT when aux_capacity is smaller (then *num_aux says how many are needed).  The zk randomness
 * hashes (:268-276) are the caller's own salts at the sorted positions. */
int iopx_merkle_membership_proof_dev(const uint8_t *d_nodes, size_t num_leaves, const size_t *positions, size_t num_positions,
                                     uint8_t *aux_hashes, size_t aux_capacity, size_t *num_aux);
/* The query responses of bcs_prover::get_transcript (libiop/bcs/bcs_prover.tcc:187-197): values[p][k] = oracle_k[positions[p]]
 * (position-major), copied to host memory; oracles are n elements of elem_bytes each. */
int iopx_query_responses_dev(const void *const *d_oracles, size_t num_oracles, size_t elem_bytes, size_t n, const size_t *positions,
                             size_t num_positions, void *values);

/* ---- LDT reducer -------------------------------------------------------------------------------- */
/* combined_LDT_virtual_oracle::evaluated_contents (libiop/protocols/ldt/ldt_reducer_aux.tcc:39-131, constructor and
 * set_random_coefficients :3-37; subset_element_powers, libiop/algebra/exponentiation.tcc:3-91): the random linear
 * combination of all oracles, each submaximal one also multiplied by x^(max_degree - degree), over the whole codeword domain.
 *   d_oracles            host array of num_oracles device pointers, 2^m elements each
 *   degrees              input_oracle_degrees (host)
 *   random_coefficients  the 2 * num_oracles elements given to set_random_coefficients (host)
 * Writes 2^m elements to d_out. */
int iopx_ldt_combine_gf192_dev(const void *const *d_oracles, size_t num_oracles, const size_t *degrees,
                               const uint64_t *random_coefficients, const uint64_t *basis, size_t m, const uint64_t *shift,
                               uint64_t *d_out);
int iopx_ldt_combine_fp3_dev(const void *const *d_oracles, size_t num_oracles, const size_t *degrees,
                             const uint64_t *random_coefficients, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                             uint64_t *d_out);

/* ---- R1CS row check -------------------------------------------------------------------------------- */
/* rowcheck_ABC_virtual_oracle::evaluated_contents (libiop/protocols/encoded/common/rowcheck.tcc:16-88):
 * out[x] = Z_H(x)^-1 * (Az(x) * Bz(x) - Cz(x)) over the whole codeword domain, all buffers on the device.
 *   gf192: the constraint domain H is span(basis[0 .. constraint_dim)) + constraint_shift (a prefix of the codeword basis, as
 *          the reference assumes, :31-33)
 *   fp3:   H is the coset constraint_shift * <gen^(2^log_n / 2^constraint_log_order)> of order 2^constraint_log_order
 * Fails with IOPX_ERR_INVALID_ARGUMENT when the two domains intersect (Z_H vanishes on the codeword domain). */
/* d_out[i] = d_in[i] / Z_S(x_i) over the domain span(basis[0..m)) + shift, S = span(basis[0..sub_dim)) + sub_shift: Z_S is constant on a coset
 * of S, so this is one product per element with a per-coset inverse (the table rowcheck uses).  The pointwise form of
 * polynomial_over_vanishing_polynomial (r1cs_rs_iop.tcc:563-565) on a domain that does not meet S, when the division is exact. */
int iopx_div_by_vanishing_gf192_dev(const uint64_t *d_in, const uint64_t *basis, size_t m, const uint64_t *shift, size_t sub_dim, const uint64_t *sub_shift,
                                    uint64_t *d_out);
int iopx_rowcheck_gf192_dev(const uint64_t *d_Az, const uint64_t *d_Bz, const uint64_t *d_Cz, const uint64_t *basis, size_t m,
                            const uint64_t *shift, size_t constraint_dim, const uint64_t *constraint_shift, uint64_t *d_out);
int iopx_rowcheck_fp3_dev(const uint64_t *d_Az, const uint64_t *d_Bz, const uint64_t *d_Cz, size_t log_n, const uint64_t *gen,
                          const uint64_t *shift, size_t constraint_log_order, const uint64_t *constraint_shift, uint64_t *d_out);

/* fz_virtual_oracle::evaluated_contents (libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.tcc:181-222):
 * out[x] = fw(x) * Z_I(x) + f_1v(x), I = the input variable domain (gf192: span(input_basis) + input_shift; fp3: the coset
 * input_shift * <order 2^input_log_order subgroup>), f_1v already extended to the codeword domain (:207-212 are two ordinary
 * transforms: iopx_add_ifft_gf192 on |I| points, iopx_add_fft_gf192_dev onto the codeword domain). */
int iopx_fz_gf192_dev(const uint64_t *d_fw, const uint64_t *d_f1v, const uint64_t *basis, size_t m, const uint64_t *shift,
                      const uint64_t *input_basis, size_t input_dim, const uint64_t *input_shift, uint64_t *d_out);
int iopx_fz_fp3_dev(const uint64_t *d_fw, const uint64_t *d_f1v, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                    size_t input_log_order, const uint64_t *input_shift, uint64_t *d_out);

/* sumcheck_g_oracle::evaluated_contents (libiop/protocols/encoded/sumcheck/sumcheck.tcc:58-119; sumcheck_aux.tcc:3-32), H = the
 * summation domain, mu = claimed_sum:
 *   gf192: out[x] = f(x) - eps^-1 mu x^(|H| - 1) - Z_H(x) h(x), eps = the linear coefficient of Z_H (a zero domain element
 *          contributes 0 to the middle term, as the reference's batch inversion does)
 *   fp3:   out[x] = (f(x) - |H|^-1 mu - Z_H(x) h(x)) / x */
int iopx_sumcheck_g_gf192_dev(const uint64_t *d_f, const uint64_t *d_h, const uint64_t *basis, size_t m, const uint64_t *shift,
                              const uint64_t *summation_basis, size_t summation_dim, const uint64_t *summation_shift,
                              const uint64_t *claimed_sum, uint64_t *d_out);
int iopx_sumcheck_g_fp3_dev(const uint64_t *d_f, const uint64_t *d_h, size_t log_n, const uint64_t *gen, const uint64_t *shift,
                            size_t summation_log_order, const uint64_t *summation_shift, const uint64_t *claimed_sum, uint64_t *d_out);

/* multi_lincheck_virtual_oracle::evaluated_contents (libiop/protocols/encoded/lincheck/basic_lincheck_aux.tcc:102-144):
 * out[x] = (sum_m r_Mz[m] * Mz_m(x)) * p_alpha_prime(x) - fz(x) * p_alpha_ABC(x) over n positions; the two p_alpha codewords are the
 * ordinary transforms of :112-118 (iopx_add_fft_gf192_dev / iopx_mul_fft_fp3_dev of the polynomials set_challenge computed).
 * d_Mz: host array of num_matrices (<= 8) device pointers; r_Mz: host coefficients. */
int iopx_lincheck_gf192_dev(const uint64_t *d_fz, const void *const *d_Mz, size_t num_matrices, const uint64_t *r_Mz,
                            const uint64_t *d_p_alpha_prime, const uint64_t *d_p_alpha_ABC, size_t n, uint64_t *d_out);
int iopx_lincheck_fp3_dev(const uint64_t *d_fz, const void *const *d_Mz, size_t num_matrices, const uint64_t *r_Mz,
                          const uint64_t *d_p_alpha_prime, const uint64_t *d_p_alpha_ABC, size_t n, uint64_t *d_out);

/* ---- encoded Aurora prover: the vector-sized steps between the transforms ---------------------------- */
/* Sparse matrix x vector in CSR form (row_ptr: rows + 1 offsets, col: column of each entry, coeff: its field element), all on the
 * device: out[r] (+)= scale * sum_t coeff[t] * vec[col[t]].  Replaces r1cs_constraint_system::create_Az_Bz_Cz_from_variable_assignment
 * (libiop/relations/r1cs.tcc:236-268; vec = (1, primary, auxiliary)) and the p_alpha_ABC accumulation of
 * multi_lincheck_virtual_oracle::set_challenge (libiop/protocols/encoded/lincheck/basic_lincheck_aux.tcc:64-88: the transposed
 * matrix with rows already placed at their summation-domain index, vec = the alpha powers, scale = r_Mz[m], accumulate over m).
 * scale: NULL or one host element. */
int iopx_spmv_gf192_dev(const uint64_t *d_row_ptr, const uint32_t *d_col, const uint64_t *d_coeff, size_t rows, const uint64_t *d_vec,
                        const uint64_t *scale, int accumulate, uint64_t *d_out);
int iopx_spmv_fp3_dev(const uint64_t *d_row_ptr, const uint32_t *d_col, const uint64_t *d_coeff, size_t rows, const uint64_t *d_vec,
                      const uint64_t *scale, int accumulate, uint64_t *d_out);
/* polynomial_over_vanishing_polynomial(P, Z).first (libiop/algebra/polynomials/vanishing_polynomial.tcc:314-371,
 * linearized_polynomial.tcc:238-289): the quotient of the n_coeffs-coefficient polynomial by the vanishing polynomial of the affine
 * subspace (basis[dim], shift) / of the coset shift * <order 2^log_order>; writes n_coeffs - |domain| coefficients (nothing when
 * n_coeffs <= |domain|).  Used for f_w = f_w' / Z_I (r1cs_rs_iop.tcc:563-565) and the sumcheck's h (sumcheck.tcc:359-365). */
int iopx_poly_div_vanishing_gf192_dev(const uint64_t *d_poly, size_t n_coeffs, const uint64_t *basis, size_t dim, const uint64_t *shift,
                                      uint64_t *d_quotient);
int iopx_poly_div_vanishing_fp3_dev(const uint64_t *d_poly, size_t n_coeffs, size_t log_order, const uint64_t *shift, uint64_t *d_quotient);
/* random_linear_combination_oracle::evaluated_contents (libiop/protocols/encoded/common/random_linear_combination.tcc:27-57):
 * out[x] = sum_i coefficients[i] * oracle_i[x]; d_oracles is a host array of num_oracles (<= 16) device pointers. */
int iopx_lincomb_gf192_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, size_t n, uint64_t *d_out);
int iopx_lincomb_fp3_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, size_t n, uint64_t *d_out);
/* Elementwise helpers on device vectors (the synthetic instance of libiop/relations/examples/r1cs_examples.tcc:40-64, and
 * f_w' = z - f_1v over the variable domain, r1cs_rs_iop.tcc:406-430): sum / difference, product, inverse (zero stays zero),
 * d_out[l] = init * base^l (the alpha powers of basic_lincheck_aux.tcc:37-45). */
int iopx_gf192_add_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count);
int iopx_gf192_inv_dev(const uint64_t *d_a, uint64_t *d_out, size_t count);
int iopx_fp3_mul_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count);
int iopx_fp3_sub_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count);
int iopx_fp3_inv_dev(const uint64_t *d_a, uint64_t *d_out, size_t count);
int iopx_fp3_pow_table_dev(uint64_t *d_out, size_t count, const uint64_t *base, const uint64_t *init);

/* ---- holographic (Fractal) prover: the vector-sized steps Aurora's entry points do not cover ---------- */
/* sum_i coefficients[i] * oracle_i[x] + constant: single_matrix_denominator::evaluated_contents
 * (libiop/protocols/encoded/lincheck/holographic_lincheck_aux.tcc:117-143) with (row, col, row*col) and the coefficients
 * (-col_query, -row_query, 1), constant row_query * col_query. */
int iopx_lincomb_affine_gf192_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, const uint64_t *constant, size_t n,
                                  uint64_t *d_out);
int iopx_lincomb_affine_fp3_dev(const void *const *d_oracles, size_t num_oracles, const uint64_t *coefficients, const uint64_t *constant, size_t n,
                                uint64_t *d_out);
/* Elementwise quotient d_out[l] = d_num[l] / d_den[l] by batch inversion (libiop/algebra/utils.tcc:57-118 batch_inverse /
 * batch_inverse_and_mul); d_num NULL: plain inverses; a zero denominator yields zero.  d_out must not alias an input.
 * Callers: lagrange_polynomial::evaluations_over_field_subset (algebra/polynomials/lagrange_polynomial.tcc:66-136),
 * single_boundary_constraint::evaluated_contents (protocols/encoded/common/boundary_constraint.tcc:22-63),
 * rational_linear_combination::evaluated_contents (common/rational_linear_combination.tcc:183-209). */
int iopx_gf192_div_dev(const uint64_t *d_num, const uint64_t *d_den, uint64_t *d_out, size_t count);
int iopx_fp3_div_dev(const uint64_t *d_num, const uint64_t *d_den, uint64_t *d_out, size_t count);
/* d_out[j] = point - x_j over the whole domain (affine subspace basis[m] + shift / coset shift * <gen> of order 2^log_n): the
 * denominators x - y of lagrange_polynomial.tcc:72-87 and the shifted elements of boundary_constraint.tcc:31-47 (negated). */
int iopx_domain_offsets_gf192_dev(const uint64_t *basis, size_t m, const uint64_t *shift, const uint64_t *point, uint64_t *d_out);
int iopx_domain_offsets_fp3_dev(size_t log_n, const uint64_t *gen, const uint64_t *shift, const uint64_t *point, uint64_t *d_out);
/* d_out[j] = constant - Z_S(x_j) over the whole domain, Z_S the vanishing polynomial of the subspace vanishing_basis[vanishing_dim] +
 * vanishing_shift / of the coset vanishing_shift * <order 2^vanishing_log_order> (vanishing_polynomial::evaluations_over_field_subset,
 * libiop/algebra/polynomials/vanishing_polynomial.tcc:97-137): with constant = Z_S(alpha) the numerator of lagrange_polynomial.tcc:124-131. */
int iopx_vanishing_evals_gf192_dev(const uint64_t *basis, size_t m, const uint64_t *shift, const uint64_t *vanishing_basis, size_t vanishing_dim,
                                   const uint64_t *vanishing_shift, const uint64_t *constant, uint64_t *d_out);
int iopx_vanishing_evals_fp3_dev(size_t log_n, const uint64_t *gen, const uint64_t *shift, size_t vanishing_log_order, const uint64_t *vanishing_shift,
                                 const uint64_t *constant, uint64_t *d_out);
/* combined_numerator / combined_denominator::evaluated_contents (libiop/protocols/encoded/common/rational_linear_combination.tcc:13-108):
 * N[x] = sum_i coefficients[i] N_i[x] prod_{k != i} D_k[x], D[x] = prod_k D_k[x] for num_rationals (<= 4) rationals; host arrays of
 * device pointers. */
int iopx_rational_combine_gf192_dev(const void *const *d_numerators, const void *const *d_denominators, size_t num_rationals, const uint64_t *coefficients,
                                    size_t n, uint64_t *d_numerator_out, uint64_t *d_denominator_out);
int iopx_rational_combine_fp3_dev(const void *const *d_numerators, const void *const *d_denominators, size_t num_rationals, const uint64_t *coefficients,
                                  size_t n, uint64_t *d_numerator_out, uint64_t *d_denominator_out);
/* sumcheck_constraint_oracle::evaluated_contents (libiop/protocols/encoded/sumcheck/rational_sumcheck.tcc:58-112) over the whole
 * codeword domain, K the summation (index) domain:
 *   subspaces: (D (p + eps^-1 mu x^(|K| - 1)) - N) / Z_K, K = span(basis[0..summation_dim)) + summation_shift, d_xinv = 1 / x over the
 *              codeword domain (iopx_gf192_div_dev of iopx_domain_offsets_gf192_dev with point 0);
 *   cosets:    (D (x p + mu / |K|) - N) / Z_K, K = summation_shift * <order 2^summation_log_order>. */
int iopx_rational_sumcheck_constraint_gf192_dev(const uint64_t *d_p, const uint64_t *d_N, const uint64_t *d_D, const uint64_t *d_xinv, const uint64_t *basis,
                                                size_t m, const uint64_t *shift, size_t summation_dim, const uint64_t *summation_shift,
                                                const uint64_t *claimed_sum, uint64_t *d_out);
int iopx_rational_sumcheck_constraint_fp3_dev(const uint64_t *d_p, const uint64_t *d_N, const uint64_t *d_D, size_t log_n, const uint64_t *gen,
                                              const uint64_t *shift, size_t summation_log_order, const uint64_t *summation_shift,
                                              const uint64_t *claimed_sum, uint64_t *d_out);
/* Host helpers (no device work): Z_S(x) and Z_S's linear coefficient — its formal derivative, vanishing_polynomial.tcc:55-74 — for
 * S = span(basis[dim]) + shift (either output may be NULL); the inverse of one element. */
int iopx_gf192_vanishing_host(const uint64_t *basis, size_t dim, const uint64_t *shift, const uint64_t *x, uint64_t *value_out,
                              uint64_t *linear_coefficient_out);
int iopx_gf192_inverse_host(const uint64_t *x, uint64_t *out);

/* ---- proof of work ------------------------------------------------------------------------------ */
/* pow<FieldT, binary_hash_digest>::solve_pow (libiop/bcs/pow.tcc:67-103) with the BLAKE2b two-to-one hash: returns the
 * FIRST candidate in the reference's order (the challenge itself, then the challenge with its last 8-byte word set to
 * 0, 1, 2, ...) for which the last word of H(challenge || candidate) has its low pow_bitlen bits zero
 * (verify_pow_internal, :143-162; pow_bitlen = pow_parameters::pow_bitlen(), :21-32).  challenge, pow: 32 host bytes. */
int iopx_pow_solve_blake2b(const uint8_t *challenge, size_t pow_bitlen, uint8_t *pow);
/* The same search over one range of candidates, for a grind split across GPUs by candidate range (libiop_amd/dist.py: the smallest
 * passing index over all ranks is the reference's first hit).  Candidate index 0 is the challenge itself, index i >= 1 the challenge
 * with its last word set to i - 1; *found receives the smallest passing index in [first, first + count) or UINT64_MAX.
 * iopx_pow_candidate_blake2b writes the 32-byte answer of a candidate index (host only). */
int iopx_pow_search_blake2b(const uint8_t *challenge, size_t pow_bitlen, uint64_t first, uint64_t count, uint64_t *found);
/* ... in two halves: _begin enqueues the batch on the library's stream and returns, _end waits for it and reads the result.  Between the two the
 * caller may do host work and enqueue further launches behind the grind — the BCS prover derives, gathers and reads back its query answers there
 * (they do not depend on the proof-of-work answer, bcs_prover.tcc:52-59).  One pending search at a time. */
int iopx_pow_search_blake2b_begin(const uint8_t *challenge, size_t pow_bitlen, uint64_t first, uint64_t count);
int iopx_pow_search_blake2b_end(uint64_t *found);
int iopx_pow_candidate_blake2b(const uint8_t *challenge, uint64_t index, uint8_t *pow);
/* pow<FieldT, FieldT>::solve_pow (pow.tcc:73-84,129-141) with the Poseidon two-to-one hash over alt_bn128 Fr: the
 * smallest k >= 0 such that word 0 of two_to_one(challenge, FieldT(k)) (canonical integer) has its low pow_bitlen bits
 * zero.  challenge, pow: 4 host words each, Montgomery form. */
int iopx_pow_solve_poseidon_bn128(const iopx_poseidon_params *params, const uint64_t *challenge, size_t pow_bitlen,
                                  uint64_t *pow);

/* ---- measurement hooks -------------------------------------------------------------------------- */
/* Per-kernel timing with HIP events recorded on the library's stream around every kernel launch.
 * iopx_profile_begin() starts recording; iopx_profile_report() synchronises, stops recording and writes
 * one text line per kernel: "<kernel> <launches> <total_ms> <algorithmic_bytes> <field_products>\n" (used by bench.py for the roofline line;
 * algorithmic_bytes = elements swept x 24 x (read + write) summed over the launches, 0 for kernels that do not report it). */
/* Options: named integers that select a prover schedule or a tile geometry (the names are listed in DESIGN.md).  A name's value is what
 * iopx_set_option gave it, else the environment variable of that name (read once per process, at the name's first lookup), else the built-in
 * default.  Schedule options (IOPX_HEAD_EVAL, IOPX_MERKLE_STREAM, IOPX_DEFER_ROOTS) are looked up per proof; tile geometries are latched by
 * their component at its first use. */
int iopx_set_option(const char *name, int value);
int iopx_clear_option(const char *name);
int iopx_get_option(const char *name, int dflt);

/* One-time host-side costs since the last reset, by label — pool growth (hipMalloc), pinned staging, FFT plans and their tables, per-domain
 * tables, matrix transpositions: "<label> <count> <total ms>\n" per label.  What a process's first proof pays beyond its kernels; after
 * iopx_aurora_instance_warm / iopx_fractal_index the first proof adds (nearly) nothing to it. */
int iopx_cold_stats(char *buf, size_t cap, int reset);
int iopx_cold_add(const char *label, double ms);
int iopx_profile_begin(void);
int iopx_profile_report(char *buf, size_t cap);

/* Elementwise GF(2^192) product on the device (d_out[i] = d_a[i] * d_b[i]); used by the parity tests of
 * the field arithmetic itself and by the field-multiplication micro-benchmark. */
int iopx_gf192_mul_dev(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t count);
/* d_out[i] = d_a[i] * d_c[0]: the wave-uniform-multiplier path the butterfly kernels use. */
int iopx_gf192_mul_uniform_dev(const uint64_t *d_a, const uint64_t *d_c, uint64_t *d_out, size_t count);
/* Test entry for the half-wavefront comb product (one multiplier per 32 lanes: d_c2[0] for lanes 0..31 of a wavefront, d_c2[1] for lanes 32..63):
 * element i is multiplied when (i & lane_mask) != 0 and copied otherwise — a divergent branch around the product when lane_mask has bits below 6 —
 * and d_active[i / 64] receives the number of lanes a ballot placed right after the product sees active.  count: a multiple of 64. */
int iopx_gf192_mul_halves_dev(const uint64_t *d_a, const uint64_t *d_c2, uint64_t *d_out, uint32_t *d_active, size_t count, uint32_t lane_mask);

/* ---- support for a native prover built on this ABI (libiop_amd/cpp/{iop,aurora}.hpp; libiop_amd/csrc/prover_support.hip) ---- */
/* Pooled device allocations for oracles, trees and scratch vectors: blocks are recycled in the order of the library's stream (all
 * work of the library is enqueued on one stream), so freeing and re-allocating a codeword costs no hipMalloc / hipFree and needs no
 * synchronisation.  The reference's counterpart is the heap behind std::vector<FieldT> in oracle<FieldT> (libiop/iop/oracles.hpp:22-52). */
int iopx_pool_alloc(void **dptr, size_t bytes);
int iopx_pool_free(void *dptr);
/* Stream-ordered device-to-device copy / byte fill; small host-to-device upload through the library's pinned staging (returns
 * without synchronising; the host buffer may be released on return). */
int iopx_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes);
int iopx_memset_dev(void *dst_dev, int value, size_t bytes);
int iopx_upload_small(void *dst_dev, const void *src_host, size_t bytes);
/* d_dst[i] = d_src[d_index[i]] / d_dst[d_index[i]] = d_src[i] for elements of elem_bytes (a multiple of 8) bytes: a whole vector
 * moved through field_subset::reindex_by_subset (libiop/algebra/field_subset/field_subset.tcc:130-142) — basic_lincheck_aux.tcc:50-58,
 * r1cs_rs_iop.tcc:406-430 on multiplicative domains. */
int iopx_gather_dev(const void *d_src, const uint64_t *d_index, size_t count, size_t elem_bytes, void *d_dst);
int iopx_scatter_dev(const void *d_src, const uint64_t *d_index, size_t count, size_t elem_bytes, void *d_dst);
/* d_dst[i] = d_src[i * stride] for i < count: the evaluations over the sub-coset of order |D| / stride inside a vector over a
 * multiplicative coset D — the positions IFFT_of_known_degree reads (libiop/algebra/fft.tcc:435-456). */
int iopx_gather_stride_dev(const void *d_src, size_t count, size_t stride, size_t elem_bytes, void *d_dst);
/* *d_count += the number of 8-byte words at which the two device buffers differ (d_count: one uint64_t in device memory, zeroed by the
 * caller).  The provers use it to confirm that a codeword computed by two routes is one codeword (libiop_amd/cpp/aurora.hpp, FRI_protocol). */
int iopx_count_mismatch_dev(const void *d_a, const void *d_b, size_t bytes, uint64_t *d_count);
/* Bytes moved by every host<->device copy the library has made (optionally reset): what a caller checks to assert that no
 * codeword-sized vector crossed PCIe during a proof. */
int iopx_transfer_stats(uint64_t *h2d_bytes, uint64_t *d2h_bytes, int reset);
/* BLAKE2b (RFC 7693; libsodium's crypto_generichash_blake2b in the reference) on the host, for the Fiat-Shamir hashchain
 * (libiop/bcs/hashing/blake2b.tcc:10-110, blake2b.cpp:50-74): outlen 1..64, optional key (<= 64 bytes). */
int iopx_blake2b_host(uint8_t *out, size_t outlen, const void *msg, size_t msglen, const void *key, size_t keylen);
/* Host scalar arithmetic for per-proof constants (never codeword-sized data): GF(2^192) product; edwards_Fr sum, difference,
 * inverse, FieldT(uint64) (Montgomery form) and the modulus words (the rejection bound of blake2b.tcc:197-227). */
int iopx_gf192_host_mul(const uint64_t *a, const uint64_t *b, uint64_t *out);
int iopx_fp3_host_add(const uint64_t *a, const uint64_t *b, uint64_t *out);
int iopx_fp3_host_sub(const uint64_t *a, const uint64_t *b, uint64_t *out);
int iopx_fp3_host_inverse(const uint64_t *a, uint64_t *out);
int iopx_fp3_from_uint(uint64_t value, uint64_t *out);
int iopx_fp3_modulus(uint64_t *out);

/* ---- the Aurora prover itself (libiop_amd/csrc/prover_capi.hip over libiop_amd/cpp/aurora.hpp) ------------------------------------ */
/* aurora_snark_prover (libiop/snark/aurora_snark.tcc:119-146; non-zk, BLAKE2b, heuristic FRI / LDT-reducer soundness as in
 * profiling/instrument_aurora_snark.cpp:209-217) with every oracle resident in HBM.  An instance holds the R1CS (three CSR matrices over
 * columns 0 = the constant 1, j >= 1 = variable j - 1 of (primary, auxiliary): relations/r1cs.tcc:236-268), its device form, the tables a
 * proof derives from it, and the variable assignment in HBM; it is created once and proves any number of times.  The transcript is
 * returned in the canonical byte form of bcs_transformation_transcript (libiop_amd/cpp/iop.hpp serialize()) in a malloc'ed buffer the
 * caller releases with iopx_host_free. */
#define IOPX_FIELD_GF192 0
#define IOPX_FIELD_EDWARDS_FR 1
typedef struct iopx_r1cs {
    size_t num_constraints, num_variables, num_inputs;
    const uint64_t *row_ptr[3];   /* A, B, C: num_constraints + 1 offsets each */
    const uint32_t *col[3];       /* column of each entry */
    const uint64_t *coeff[3];     /* 3 words per entry */
} iopx_r1cs;
typedef struct iopx_aurora_instance iopx_aurora_instance;
/* assignment: num_variables elements (primary inputs first), host memory */
int iopx_aurora_instance_create(const iopx_r1cs *r1cs, const uint64_t *assignment, int field, iopx_aurora_instance **out);
/* generate_r1cs_example (libiop/relations/examples/r1cs_examples.tcc:23-78) seeded with SplitMix64 */
int iopx_aurora_example_instance_create(int field, size_t num_constraints, size_t num_inputs, size_t num_variables, uint64_t seed,
                                        iopx_aurora_instance **out);
/* One-time work of a process's first proof, done ahead of it: grows the device pool to a proof's footprint, builds the transforms' plans and
 * tables for every domain of this parameter set, the virtual oracles' per-domain tables and the lincheck's transposed matrices, by running the
 * prover once on the instance's own assignment and discarding the argument (protocol 0: Aurora; 1: Fractal, after iopx_fractal_index).  The
 * proofs after it take the steady-state time (bench.py: config.first_proof_ms beside config.instance_setup_ms; iopx_cold_stats shows what was built). */
int iopx_aurora_instance_warm(iopx_aurora_instance *instance, int protocol, size_t security_parameter, size_t RS_extra_dimensions, size_t FRI_localization_parameter);
int iopx_aurora_prove(iopx_aurora_instance *instance, size_t security_parameter, size_t RS_extra_dimensions, size_t FRI_localization_parameter,
                      uint8_t **transcript, size_t *transcript_bytes);
/* fractal_snark_indexer / fractal_snark_prover (libiop/snark/fractal_snark.tcc:114-162; non-zk, BLAKE2b; RS_extra_dimensions 3 and
 * localization 2 in profiling/instrument_fractal_snark.cpp:93-120) on the same instance handle (square matrices: num_constraints =
 * num_variables + 1).  iopx_fractal_index builds the prover index — twelve index oracles over the codeword domain, their Merkle tree and
 * their evaluations over the index domain, kept in HBM inside the instance — and returns the verifier index, the tree's root(s), 32 bytes
 * each; iopx_fractal_prove proves against that index any number of times (same parameters) and returns the transcript without the
 * index's roots, as bcs_prover::get_transcript does for a holographic protocol (bcs_prover.tcc:119-134). */
int iopx_fractal_index(iopx_aurora_instance *instance, size_t security_parameter, size_t RS_extra_dimensions, size_t FRI_localization_parameter,
                       uint8_t *index_roots, size_t root_capacity, size_t *num_roots);
int iopx_fractal_prove(iopx_aurora_instance *instance, size_t security_parameter, size_t RS_extra_dimensions, size_t FRI_localization_parameter,
                       uint8_t **transcript, size_t *transcript_bytes);
/* FRI_snark_prover (libiop/snark/fri_snark.tcc:43-77, protocols/fri_iop.tcc:3-101; BASELINE configs[2]): the FRI-only SNARK for the polynomial
 * with the given coefficients (device memory, at most 2^(codeword_domain_dim - RS_extra_dimensions)) over the unshifted default codeword
 * domain of 2^codeword_domain_dim points: extension, commitment, LDT reduction, FRI rounds, proof of work (codeword_domain_dim + 3 bits),
 * queries.  The repetitions are the harness's (profiling/instrument_fri_snark.cpp:84-148: 1 interactive, 10 query).  Transcript as above.
 * The _dist form distributes the codeword over `comm` like the other provers (NULL = one GPU). */
int iopx_fri_snark_prove(int field, const uint64_t *d_poly_coeffs, size_t n_coeffs, size_t codeword_domain_dim, size_t RS_extra_dimensions,
                         size_t FRI_localization_parameter, size_t num_interactive_repetitions, size_t num_query_repetitions, uint8_t **transcript,
                         size_t *transcript_bytes);
int iopx_aurora_instance_free(iopx_aurora_instance *instance);
int iopx_host_free(void *p);

/* ---- multi-GPU: one process per GPU, one communicator per process -------------------------------------
 * The reference prover is a single process (snark/aurora_snark.tcc:119-146); its counterpart over N GPUs runs the same call in N
 * processes, each holding 1/N of every codeword-domain vector: contiguous cosets of the systematic sub-domain for affine subspaces
 * (subspace.tcc:73-91: FRI cosets and Merkle leaves are contiguous runs), residue classes for multiplicative cosets (subgroup.tcc:175-197:
 * coset {j + k n/c} lies in one residue class).  What crosses GPUs is exchanged through an iopx_comm:
 *   iopx_comm_create_rccl        RCCL over xGMI, created from a 128-byte ncclUniqueId that rank 0 obtains with iopx_comm_rccl_unique_id and
 *                                hands to the other ranks by whatever the host program has (torch.distributed store, MPI, a file);
 *                                collectives take DEVICE pointers and are enqueued on the library's stream (no host synchronisation).
 *                                librccl is resolved at run time (the copy already loaded into the process when there is one).
 *   iopx_comm_create_callbacks   the host program's own transport (it already holds a communicator; or the CPU test-suite's gloo group):
 *                                every collective is forwarded to the callbacks with the pointers as given and the library's stream.
 * Every collective below is collective over all ranks of the communicator and must be called in the same order on every rank. */
typedef struct iopx_comm iopx_comm;
#define IOPX_COMM_UNIQUE_ID_BYTES 128
#define IOPX_COMM_SUM 0        /* wrapping sum of uint64 words */
#define IOPX_COMM_MIN 1        /* minimum of uint64 words */
typedef struct iopx_comm_callbacks {
    void *user;
    /* recv[r * bytes .. (r+1) * bytes) = rank r's send[0 .. bytes) */
    int (*all_gather)(void *user, const void *send, void *recv, size_t bytes_per_rank, void *hip_stream);
    /* in place over `count` uint64 words */
    int (*all_reduce_u64)(void *user, void *buf, size_t count, int op, void *hip_stream);
    int (*broadcast)(void *user, void *buf, size_t bytes, int root, void *hip_stream);
    /* recv[r * bytes ..) = rank r's send[my_rank * bytes ..) */
    int (*all_to_all)(void *user, const void *send, void *recv, size_t bytes_per_rank, void *hip_stream);
    /* symmetric exchange with one peer (the peer calls it with this rank as its peer) */
    int (*sendrecv)(void *user, const void *send, void *recv, size_t bytes, int peer, void *hip_stream);
} iopx_comm_callbacks;
int iopx_comm_rccl_unique_id(uint8_t *unique_id /* IOPX_COMM_UNIQUE_ID_BYTES */);
int iopx_comm_create_rccl(int rank, int world, const uint8_t *unique_id, iopx_comm **out);
int iopx_comm_create_callbacks(int rank, int world, const iopx_comm_callbacks *callbacks, iopx_comm **out);
/* Rank `rank` of a `world`-rank job played ALONE on this GPU, for measuring one rank's compute path where no multi-GPU node is at hand
 * (bench.py --replay-rank): every collective is completed locally on the library's stream — an all-gather fills every slot with this rank's
 * part, an all-to-all and a send/receive copy the send buffer, an all-reduce and a broadcast leave the buffer as it is — so the kernels the rank
 * would run between its collectives run, on inputs that are field elements but NOT the peers' real data: the transcript of such a proof is
 * meaningless and must not be used.  The proof-of-work search of a replayed rank covers all ranks' candidate ranges (it has nobody to hear a hit from). */
int iopx_comm_create_replay(int rank, int world, iopx_comm **out);
int iopx_comm_is_replay(const iopx_comm *comm);
int iopx_comm_destroy(iopx_comm *comm);
int iopx_comm_rank(const iopx_comm *comm, int *rank, int *world);
int iopx_comm_all_gather_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);
int iopx_comm_all_reduce_u64_dev(iopx_comm *comm, void *d_buf, size_t count, int op);
int iopx_comm_broadcast_dev(iopx_comm *comm, void *d_buf, size_t bytes, int root);
int iopx_comm_all_to_all_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);
int iopx_comm_sendrecv_dev(iopx_comm *comm, const void *d_send, void *d_recv, size_t bytes, int peer);
/* ONE additive transform as long as its domain, sharded across the N = 2^r ranks of `comm` (libiop_amd/csrc/fft_add_dist.hip): additive_FFT /
 * additive_IFFT (libiop/algebra/fft.tcc:39-204) of 2^m coefficients over a 2^m-point affine subspace, every rank holding the contiguous block
 * [rank 2^m / N, (rank + 1) 2^m / N) of the input and of the output (device memory, 2^m / N elements).  One all-to-all transposes the coefficients
 * into the layout in which the Gao-Mateer recursion's top r levels are shard-local or whole-shard exchanges between peers, a complete local
 * transform of 2^(m-r) points follows, and the last r butterfly levels exchange shards between peers.  m >= 2 r (the transpose moves 2^(m-2r)
 * elements between every pair of ranks); shorter transforms are refused with IOPX_ERR_INVALID_ARGUMENT. */
int iopx_add_fft_gf192_dist_dev(iopx_comm *comm, const uint64_t *d_block_coeffs, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *d_block_out);
int iopx_add_ifft_gf192_dist_dev(iopx_comm *comm, const uint64_t *d_block_evals, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *d_block_out);
/* While a communicator of N > 1 ranks is bound, every additive transform of the library splits its Gao-Mateer phase 1 (the twists and
 * Taylor expansions on the 2^d coefficients, fft.tcc:62-83 / :172-200) over the ranks when d >= 16: the first log2 N levels run on the
 * whole vector on every rank, from then on the sub-polynomials of different residues of the coefficient index mod N never mix, so rank
 * r runs the remaining levels on the residue class r (a contiguous copy) and one all-gather reassembles the vector.  REQUIRES that every
 * rank issues the same sequence of transforms on the same (replicated) inputs — what the distributed provers do for everything over
 * the <= 2^21-element domains; a section that only one rank executes must unbind first.  NULL unbinds.  The binding belongs to the calling
 * host thread (transforms issued from other threads are not affected); the library's stream is one per process all the same, so the prover
 * entry points of one process are meant to be driven from one thread at a time. */
int iopx_comm_bind_transforms(iopx_comm *comm);
/* collectives issued and payload bytes sent by this rank through any communicator since the last reset (reset != 0 clears them) */
int iopx_comm_stats(uint64_t *num_collectives, uint64_t *bytes, int reset);
/* d_dst[(i * parts + r) * elem_bytes ..] = d_src[(r * count + i) * elem_bytes ..]: `parts` residue classes stored back to back
 * (an all-gather's output) -> natural order */
int iopx_interleave_dev(const void *d_src, size_t parts, size_t count, size_t elem_bytes, void *d_dst);
/* d_out[(dst_row[i] * num_srcs + k) * elem_bytes ..] = d_srcs[k][src_index[i] * elem_bytes ..] for i < count: the rows of a query answer
 * or an authentication path this rank owns, placed into a zeroed buffer that an all-reduce completes (every row has one owner).
 * src_index, dst_row: host arrays. */
int iopx_gather_rows_dev(const void *const *d_srcs, size_t num_srcs, size_t elem_bytes, const uint64_t *src_index, const uint64_t *dst_row,
                         size_t count, void *d_out);
/* The provers with every codeword-domain vector distributed over the ranks of `comm` (libiop_amd/cpp/dist.hpp): every rank calls with
 * the same instance and parameters and receives the same transcript, byte-identical to the single-GPU call's.  comm == NULL or a
 * one-rank communicator is the single-GPU prover (through the same code path).  iopx_fractal_index_dist leaves this rank's part of
 * the index inside the instance. */
int iopx_aurora_prove_dist(iopx_aurora_instance *instance, iopx_comm *comm, size_t security_parameter, size_t RS_extra_dimensions,
                           size_t FRI_localization_parameter, uint8_t **transcript, size_t *transcript_bytes);
int iopx_fractal_index_dist(iopx_aurora_instance *instance, iopx_comm *comm, size_t security_parameter, size_t RS_extra_dimensions,
                            size_t FRI_localization_parameter, uint8_t *index_roots, size_t root_capacity, size_t *num_roots);
int iopx_fractal_prove_dist(iopx_aurora_instance *instance, iopx_comm *comm, size_t security_parameter, size_t RS_extra_dimensions,
                            size_t FRI_localization_parameter, uint8_t **transcript, size_t *transcript_bytes);
int iopx_fri_snark_prove_dist(int field, iopx_comm *comm, const uint64_t *d_poly_coeffs, size_t n_coeffs, size_t codeword_domain_dim, size_t RS_extra_dimensions,
                              size_t FRI_localization_parameter, size_t num_interactive_repetitions, size_t num_query_repetitions, uint8_t **transcript,
                              size_t *transcript_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LIBIOP_AMD_H */
