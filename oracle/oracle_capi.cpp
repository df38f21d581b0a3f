// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
// extern "C" surface over the templated restatement, for ctypes (tests/, smoke(), bench cpu_baseline).
#include <cstdint>
#include <cstring>
#include <vector>
#include "field.hpp"
#include "algebra.hpp"
#include "fri.hpp"
#include "merkle.hpp"
#include "mult.hpp"
#include "poseidon.hpp"
#include "pow.hpp"
#include "ldt.hpp"
#include "aurora.hpp"
#include "fractal.hpp"

using namespace oracle;

namespace {

template<typename F>
std::vector<F> load(const uint64_t *p, size_t count)
{
    std::vector<F> v(count);
    if (count) memcpy((void *)v.data(), p, count * sizeof(F));
    return v;
}
template<typename F>
void store(uint64_t *p, const std::vector<F> &v) { if (!v.empty()) memcpy(p, (const void *)v.data(), v.size() * sizeof(F)); }

template<typename F>
affine_subspace<F> load_domain(const uint64_t *basis, size_t m, const uint64_t *shift)
{
    F s; memcpy((void *)&s, shift, sizeof(F));
    return affine_subspace<F>(load<F>(basis, m), s);
}

#define DISPATCH(words, CALL)                         \
    switch (words) {                                  \
    case 1: { typedef gf64 F; CALL; } break;          \
    case 2: { typedef gf128 F; CALL; } break;         \
    case 3: { typedef gf192 F; CALL; } break;         \
    case 4: { typedef gf256 F; CALL; } break;         \
    default: return -1;                               \
    }

} // namespace

extern "C" {

int oracle_has_pclmul(void)
{
#if defined(__PCLMUL__)
    return 1;
#else
    return 0;
#endif
}

void oracle_clmul64(uint64_t a, uint64_t b, uint64_t *lo, uint64_t *hi) { clmul64(a, b, *lo, *hi); }
void oracle_clmul64_portable(uint64_t a, uint64_t b, uint64_t *lo, uint64_t *hi) { clmul64_portable(a, b, *lo, *hi); }

int oracle_gf_mul(int words, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t count)
{
    DISPATCH(words, {
        const F *x = (const F *)a; const F *y = (const F *)b; F *o = (F *)out;
        for (size_t i = 0; i < count; ++i) o[i] = x[i] * y[i];
    });
    return 0;
}

int oracle_gf_inv(int words, const uint64_t *a, uint64_t *out, size_t count)
{
    DISPATCH(words, {
        const F *x = (const F *)a; F *o = (F *)out;
        for (size_t i = 0; i < count; ++i) o[i] = x[i].inverse();
    });
    return 0;
}

int oracle_all_subset_sums(int words, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *out)
{
    DISPATCH(words, { store<F>(out, load_domain<F>(basis, m, shift).all_elements()); });
    return 0;
}

int oracle_naive_fft(int words, const uint64_t *coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                     const uint64_t *shift, uint64_t *out)
{
    DISPATCH(words, {
        const affine_subspace<F> d = load_domain<F>(basis, m, shift);
        store<F>(out, naive_FFT<F>(load<F>(coeffs, n_coeffs), d.all_elements()));
    });
    return 0;
}

int oracle_additive_fft(int words, const uint64_t *coeffs, size_t n_coeffs, const uint64_t *basis, size_t m,
                        const uint64_t *shift, uint64_t *out)
{
    if (n_coeffs > ((size_t)1 << m)) return -2;
    DISPATCH(words, { store<F>(out, additive_FFT<F>(load<F>(coeffs, n_coeffs), load_domain<F>(basis, m, shift))); });
    return 0;
}

int oracle_additive_ifft(int words, const uint64_t *evals, const uint64_t *basis, size_t m, const uint64_t *shift,
                         uint64_t *out)
{
    DISPATCH(words, { store<F>(out, additive_IFFT<F>(load<F>(evals, (size_t)1 << m), load_domain<F>(basis, m, shift))); });
    return 0;
}

// out receives 2^ceil(log2 degree) coefficients
int oracle_additive_ifft_known_degree(int words, const uint64_t *evals, size_t degree, const uint64_t *basis, size_t m,
                                      const uint64_t *shift, uint64_t *out)
{
    DISPATCH(words, {
        store<F>(out, additive_IFFT_of_known_degree<F>(load<F>(evals, (size_t)1 << m), degree, load_domain<F>(basis, m, shift)));
    });
    return 0;
}

int oracle_fri_fold_additive(int words, const uint64_t *f_i, const uint64_t *basis, size_t m, const uint64_t *shift,
                             size_t coset_size, const uint64_t *x_i, uint64_t *out)
{
    DISPATCH(words, {
        F x; memcpy((void *)&x, x_i, sizeof(F));
        store<F>(out, additive_evaluate_next_f_i_over_entire_domain<F>(load<F>(f_i, (size_t)1 << m),
                                                                      load_domain<F>(basis, m, shift), coset_size, x));
    });
    return 0;
}

// Domain chain: out_bases holds the bases of L^(1), L^(2), ... concatenated (dims m - eta_0, ...),
// out_shifts one shift per derived domain.
int oracle_fri_domains_additive(int words, const uint64_t *basis, size_t m, const uint64_t *shift,
                                const size_t *loc, size_t num_loc, uint64_t *out_bases, uint64_t *out_shifts)
{
    DISPATCH(words, {
        const std::vector<affine_subspace<F>> doms =
            fri_additive_domains<F>(load_domain<F>(basis, m, shift), std::vector<size_t>(loc, loc + num_loc));
        size_t off = 0;
        for (size_t i = 1; i < doms.size(); ++i) {
            memcpy(out_bases + off, (const void *)doms[i].basis.data(), doms[i].basis.size() * sizeof(F));
            off += doms[i].basis.size() * (sizeof(F) / 8);
            memcpy(out_shifts + (i - 1) * (sizeof(F) / 8), (const void *)&doms[i].shift, sizeof(F));
        }
    });
    return 0;
}

size_t oracle_localization_array(size_t loc_param, size_t codeword_dim, size_t rs_extra, size_t *out, size_t cap)
{
    const std::vector<size_t> v = localization_parameter_to_array(loc_param, codeword_dim, rs_extra);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}

void oracle_next_coset_query_positions(int additive, size_t non_localized_n, size_t localized_n, size_t seed_position,
                                       size_t prev_loc, size_t cur_loc, size_t *out)
{
    const std::vector<size_t> v = next_coset_query_positions(additive != 0, non_localized_n, localized_n, seed_position, prev_loc, cur_loc);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
}

void oracle_blake2b(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen, const uint8_t *key, size_t keylen)
{
    blake2b(out, outlen, in, inlen, key, keylen);
}

int oracle_merkle_build(const uint8_t *const *oracles, size_t num_oracles, size_t elem_bytes, size_t n, size_t coset_size,
                        int additive, const uint8_t *salts, size_t salt_bytes, uint8_t *nodes)
{
    try {
        merkle_build(oracles, num_oracles, elem_bytes, n, coset_size, additive != 0, salts, salt_bytes, nodes);
    } catch (const std::invalid_argument &) { return -3; }
    return 0;
}

// hashchain, functional form: state (32 B) and squeeze index are caller-held
void oracle_hashchain_init(uint8_t *state, uint64_t *squeeze_index)
{
    blake2b_hashchain hc; memcpy(state, hc.state, DIGEST_LEN); *squeeze_index = 0;
}
void oracle_hashchain_absorb(uint8_t *state, const uint8_t *digest)
{
    blake2b_hashchain hc; memcpy(hc.state, state, DIGEST_LEN); hc.absorb_digest(digest); memcpy(state, hc.state, DIGEST_LEN);
}
void oracle_hashchain_squeeze_binary(const uint8_t *state, uint64_t *squeeze_index, size_t num_elements, size_t elem_bytes, uint8_t *out)
{
    blake2b_hashchain hc; memcpy(hc.state, state, DIGEST_LEN); hc.squeeze_index = *squeeze_index;
    hc.squeeze_binary_field(num_elements, elem_bytes, out); *squeeze_index = hc.squeeze_index;
}
int oracle_hashchain_squeeze_positions(const uint8_t *state, uint64_t *squeeze_index, size_t num_positions, size_t range, size_t *out)
{
    blake2b_hashchain hc; memcpy(hc.state, state, DIGEST_LEN); hc.squeeze_index = *squeeze_index;
    try {
        const std::vector<size_t> v = hc.squeeze_query_positions(num_positions, range);
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    } catch (const std::invalid_argument &) { return -3; }
    *squeeze_index = hc.squeeze_index;
    return 0;
}


// ---- prime field edwards_Fr (3 limbs, Montgomery words) and the multiplicative-domain path ----------------
typedef edwards_Fr FP;

void oracle_fp_from_canonical(const uint64_t *c, uint64_t *out, size_t count)
{
    for (size_t i = 0; i < count; ++i) { const FP x = FP::from_canonical(c + 3 * i); memcpy(out + 3 * i, x.mont, 24); }
}
void oracle_fp_to_canonical(const uint64_t *m, uint64_t *out, size_t count)
{
    for (size_t i = 0; i < count; ++i) { FP x; memcpy(x.mont, m + 3 * i, 24); x.to_canonical(out + 3 * i); }
}
void oracle_fp_binop(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t count)
{
    const FP *x = (const FP *)a; const FP *y = (const FP *)b; FP *o = (FP *)out;
    for (size_t i = 0; i < count; ++i) o[i] = op == 0 ? x[i] * y[i] : (op == 1 ? x[i] + y[i] : x[i] - y[i]);
}
void oracle_fp_inv(const uint64_t *a, uint64_t *out, size_t count)
{
    const FP *x = (const FP *)a; FP *o = (FP *)out;
    for (size_t i = 0; i < count; ++i) o[i] = x[i].inverse();
}
void oracle_fp_subgroup_generator(size_t order, uint64_t *out) { const FP g = FP::subgroup_generator(order); memcpy(out, g.mont, 24); }

static mult_coset<FP> load_coset(size_t order, const uint64_t *shift)
{
    FP s; memcpy(s.mont, shift, 24);
    return mult_coset<FP>(order, s);
}
void oracle_fp_all_elements(size_t order, const uint64_t *shift, uint64_t *out) { store<FP>(out, load_coset(order, shift).all_elements()); }
void oracle_fp_naive_fft(const uint64_t *coeffs, size_t n_coeffs, size_t order, const uint64_t *shift, uint64_t *out)
{
    store<FP>(out, naive_FFT<FP>(load<FP>(coeffs, n_coeffs), load_coset(order, shift).all_elements()));
}
int oracle_fp_fft(const uint64_t *coeffs, size_t n_coeffs, size_t order, const uint64_t *shift, uint64_t *out)
{
    if (n_coeffs > order || n_coeffs == 0) return -2;
    store<FP>(out, multiplicative_FFT_degree_aware<FP>(load<FP>(coeffs, n_coeffs), load_coset(order, shift)));
    return 0;
}
void oracle_fp_ifft(const uint64_t *evals, size_t order, const uint64_t *shift, uint64_t *out)
{
    store<FP>(out, multiplicative_IFFT<FP>(load<FP>(evals, order), load_coset(order, shift)));
}
void oracle_fp_ifft_known_degree(const uint64_t *evals, size_t degree, size_t order, const uint64_t *shift, uint64_t *out)
{
    store<FP>(out, multiplicative_IFFT_of_known_degree<FP>(load<FP>(evals, order), degree, load_coset(order, shift)));
}
void oracle_fp_fri_fold(const uint64_t *f_i, size_t order, const uint64_t *shift, size_t coset_size, const uint64_t *x_i, uint64_t *out)
{
    FP x; memcpy(x.mont, x_i, 24);
    store<FP>(out, multiplicative_evaluate_next_f_i_over_entire_domain<FP>(load<FP>(f_i, order), load_coset(order, shift), coset_size, x));
}


// ---- Poseidon over alt_bn128 Fr (4 limbs).  Parameters travel as canonical 4-word integers. -------------------
typedef alt_bn128_Fr BN;

static poseidon_params<BN> load_poseidon(size_t alpha, size_t full_rounds, size_t partial_rounds, size_t rate, size_t t,
                                         int near_mds, const uint64_t *mds, const uint64_t *ark)
{
    poseidon_params<BN> p;
    p.alpha = alpha; p.full_rounds = full_rounds; p.partial_rounds = partial_rounds; p.rate = rate; p.state_size = t; p.near_mds = near_mds != 0;
    for (size_t r = 0; r < t; ++r) {
        std::vector<BN> row;
        for (size_t c = 0; c < t; ++c) row.push_back(BN::from_canonical(mds + 4 * (r * t + c)));
        p.mds.push_back(row);
    }
    for (size_t r = 0; r < full_rounds + partial_rounds; ++r) {
        std::vector<BN> row;
        for (size_t c = 0; c < t; ++c) row.push_back(BN::from_canonical(ark + 4 * (r * t + c)));
        p.ark.push_back(row);
    }
    return p;
}

void oracle_bn_from_canonical(const uint64_t *c, uint64_t *out, size_t count)
{
    for (size_t i = 0; i < count; ++i) { const BN x = BN::from_canonical(c + 4 * i); memcpy(out + 4 * i, x.mont, 32); }
}
void oracle_bn_to_canonical(const uint64_t *m, uint64_t *out, size_t count)
{
    for (size_t i = 0; i < count; ++i) { BN x; memcpy(x.mont, m + 4 * i, 32); x.to_canonical(out + 4 * i); }
}

// state (t Montgomery elements) <- permutation(state)
void oracle_poseidon_permute(size_t alpha, size_t fr, size_t pr, size_t rate, size_t t, int near_mds, const uint64_t *mds,
                             const uint64_t *ark, uint64_t *state)
{
    poseidon_sponge<BN> sp(load_poseidon(alpha, fr, pr, rate, t, near_mds, mds, ark));
    memcpy((void *)sp.state.data(), state, t * 32);
    sp.permute();
    memcpy(state, (const void *)sp.state.data(), t * 32);
}

// out = algebraic_leafhash::hash(leaf[count]) ; Montgomery words in / out
void oracle_poseidon_leafhash(size_t alpha, size_t fr, size_t pr, size_t rate, size_t t, int near_mds, const uint64_t *mds,
                              const uint64_t *ark, const uint64_t *leaf, size_t count, uint64_t *out)
{
    const BN h = poseidon_leafhash<BN>(load_poseidon(alpha, fr, pr, rate, t, near_mds, mds, ark), load<BN>(leaf, count));
    memcpy(out, h.mont, 32);
}

void oracle_poseidon_two_to_one(size_t alpha, size_t fr, size_t pr, size_t rate, size_t t, int near_mds, const uint64_t *mds,
                                const uint64_t *ark, const uint64_t *l, const uint64_t *r, uint64_t *out)
{
    BN a, b; memcpy(a.mont, l, 32); memcpy(b.mont, r, 32);
    const BN h = poseidon_two_to_one<BN>(load_poseidon(alpha, fr, pr, rate, t, near_mds, mds, ark), a, b);
    memcpy(out, h.mont, 32);
}

void oracle_poseidon_salt_to_field(const uint8_t *salt, uint64_t *out)
{
    const BN x = poseidon_salt_to_field<BN>(salt);
    memcpy(out, x.mont, 32);
}

// Merkle tree with algebraic hashes (merkle_tree.tcc:92-229 with algebraic_leafhash / algebraic_two_to_one_hash):
// nodes = (2L - 1) field elements (Montgomery words), heap order.
void oracle_poseidon_merkle(size_t alpha, size_t fr, size_t pr, size_t rate, size_t t, int near_mds, const uint64_t *mds,
                            const uint64_t *ark, const uint64_t *const *oracles, size_t num_oracles, size_t n, size_t coset_size,
                            int additive, const uint8_t *salts, uint64_t *nodes)
{
    const poseidon_params<BN> P = load_poseidon(alpha, fr, pr, rate, t, near_mds, mds, ark);
    const size_t L = n / coset_size;
    std::vector<BN> nd(2 * L - 1);
    for (size_t i = 0; i < L; ++i) {
        std::vector<BN> slice(num_oracles * coset_size);
        for (size_t j = 0; j < coset_size; ++j) {
            const size_t pos = position_by_coset_indices(additive != 0, n, i, j, coset_size);
            for (size_t k = 0; k < num_oracles; ++k) memcpy(slice[j + k * coset_size].mont, oracles[k] + 4 * pos, 32);
        }
        if (salts) slice.push_back(poseidon_salt_to_field<BN>(salts + 32 * i));     // zk_hash, algebraic_sponge.tcc:232-245
        nd[L - 1 + i] = poseidon_leafhash<BN>(P, slice);
    }
    for (size_t j = L - 1; j-- > 0; ) nd[j] = poseidon_two_to_one<BN>(P, nd[2 * j + 1], nd[2 * j + 2]);
    memcpy(nodes, (const void *)nd.data(), nd.size() * 32);
}

// proof of work (pow.tcc:21-32,73-162)
size_t oracle_pow_bitlen(size_t work_parameter, size_t cost_per_hash) { return pow_bitlen(work_parameter, cost_per_hash); }
int oracle_pow_verify_blake2b(const uint8_t *challenge, const uint8_t *pow, size_t bitlen) { return pow_verify_blake2b(challenge, pow, bitlen) ? 1 : 0; }
uint64_t oracle_pow_solve_blake2b(const uint8_t *challenge, size_t bitlen, uint8_t *pow) { return pow_solve_blake2b(challenge, bitlen, pow); }
int oracle_pow_verify_poseidon(size_t alpha, size_t fr, size_t pr, size_t rate, size_t t, int near_mds, const uint64_t *mds,
                               const uint64_t *ark, const uint64_t *challenge, const uint64_t *pow, size_t bitlen)
{
    const poseidon_params<BN> P = load_poseidon(alpha, fr, pr, rate, t, near_mds, mds, ark);
    BN c, w; memcpy(c.mont, challenge, 32); memcpy(w.mont, pow, 32);
    return pow_verify_poseidon<BN>(P, c, w, bitlen) ? 1 : 0;
}
uint64_t oracle_pow_solve_poseidon(size_t alpha, size_t fr, size_t pr, size_t rate, size_t t, int near_mds, const uint64_t *mds,
                                   const uint64_t *ark, const uint64_t *challenge, size_t bitlen, uint64_t *pow)
{
    const poseidon_params<BN> P = load_poseidon(alpha, fr, pr, rate, t, near_mds, mds, ark);
    BN c, w; memcpy(c.mont, challenge, 32);
    const uint64_t calls = pow_solve_poseidon<BN>(P, c, bitlen, w);
    memcpy(pow, w.mont, 32);
    return calls;
}

// LDT reducer (ldt_reducer_aux.tcc:3-136, exponentiation.tcc:3-91)
int oracle_subspace_element_powers(int words, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t exponent, uint64_t *out)
{
    DISPATCH(words, { store<F>(out, subspace_element_powers<F>(load_domain<F>(basis, m, shift), exponent)); });
    return 0;
}
// coefficients: the 2 * num_oracles elements handed to set_random_coefficients
int oracle_ldt_combine_additive(int words, const uint64_t *const *evals, size_t num_oracles, const size_t *degrees,
                                const uint64_t *coefficients, const uint64_t *basis, size_t m, const uint64_t *shift, uint64_t *out)
{
    DISPATCH(words, {
        combined_LDT_virtual_oracle<F> vo(std::vector<size_t>(degrees, degrees + num_oracles));
        vo.set_random_coefficients(load<F>(coefficients, 2 * num_oracles));
        std::vector<std::vector<F>> ev;
        for (size_t k = 0; k < num_oracles; ++k) ev.push_back(load<F>(evals[k], (size_t)1 << m));
        store<F>(out, vo.evaluated_contents(load_domain<F>(basis, m, shift), ev));
    });
    return 0;
}
void oracle_ldt_combine_fp(const uint64_t *const *evals, size_t num_oracles, const size_t *degrees, const uint64_t *coefficients,
                           size_t order, const uint64_t *shift, uint64_t *out)
{
    combined_LDT_virtual_oracle<FP> vo(std::vector<size_t>(degrees, degrees + num_oracles));
    vo.set_random_coefficients(load<FP>(coefficients, 2 * num_oracles));
    std::vector<std::vector<FP>> ev;
    for (size_t k = 0; k < num_oracles; ++k) ev.push_back(load<FP>(evals[k], order));
    store<FP>(out, vo.evaluated_contents(load_coset(order, shift), ev));
}
void oracle_fp_coset_element_powers(size_t order, const uint64_t *shift, uint64_t exponent, uint64_t *out)
{
    store<FP>(out, coset_element_powers<FP>(load_coset(order, shift), exponent));
}

// membership proofs (merkle_tree.tcc:242-515)
// returns the number of auxiliary node indices written to out (cap entries available), or (size_t)-1 for a bad position
size_t oracle_membership_proof_indices(size_t num_leaves, const size_t *positions, size_t count, size_t *out, size_t cap)
{
    try {
        const std::vector<size_t> v = membership_proof_node_indices(num_leaves, std::vector<size_t>(positions, positions + count));
        for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
        return v.size();
    } catch (const std::invalid_argument &) { return (size_t)-1; }
}
int oracle_membership_proof_validate(const uint8_t *root, size_t num_leaves, const size_t *positions, size_t count,
                                     const uint8_t *leaf_hashes, const uint8_t *aux, size_t num_aux)
{
    std::vector<std::vector<uint8_t>> lh, ax;
    for (size_t i = 0; i < count; ++i) lh.emplace_back(leaf_hashes + 32 * i, leaf_hashes + 32 * i + 32);
    for (size_t i = 0; i < num_aux; ++i) ax.emplace_back(aux + 32 * i, aux + 32 * i + 32);
    try {
        return membership_proof_validate(root, num_leaves, std::vector<size_t>(positions, positions + count), lh, ax) ? 1 : 0;
    } catch (const std::logic_error &) { return -1; }
}
size_t oracle_count_hashes_to_verify(size_t num_leaves, const size_t *positions, size_t count)
{
    return count_hashes_to_verify_set_membership_proof(num_leaves, std::vector<size_t>(positions, positions + count));
}

// verifier-side single-coset fold (fri_aux.tcc:270-303) and Horner evaluation of the final polynomial
int oracle_fri_fold_at_coset(int words, const uint64_t *coset_evals, size_t coset_size, const uint64_t *coset_basis, size_t eta,
                             const uint64_t *shift, const uint64_t *x_i, uint64_t *out)
{
    DISPATCH(words, {
        F s; memcpy((void *)&s, shift, sizeof(F));
        F x; memcpy((void *)&x, x_i, sizeof(F));
        const F r = additive_evaluate_next_f_i_at_coset<F>(load<F>(coset_evals, coset_size), load<F>(coset_basis, eta), s, x);
        memcpy(out, (const void *)&r, sizeof(F));
    });
    return 0;
}
int oracle_poly_eval(int words, const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out)
{
    DISPATCH(words, {
        F xx; memcpy((void *)&xx, x, sizeof(F));
        const std::vector<F> c = load<F>(coeffs, n);
        F r = F::zero();
        for (size_t i = n; i-- > 0; ) { r *= xx; r += c[i]; }
        memcpy(out, (const void *)&r, sizeof(F));
    });
    return 0;
}

// multiplicative verifier-side fold (fri_aux.tcc:305-349) and Horner evaluation over the prime field
void oracle_fp_fri_fold_at_coset(const uint64_t *coset_evals, size_t coset_size, const uint64_t *g, const uint64_t *h, const uint64_t *x_i, uint64_t *out)
{
    FP gg, hh, x; memcpy(gg.mont, g, 24); memcpy(hh.mont, h, 24); memcpy(x.mont, x_i, 24);
    const FP r = multiplicative_evaluate_next_f_i_at_coset<FP>(load<FP>(coset_evals, coset_size), gg, hh, x);
    memcpy(out, r.mont, 24);
}
void oracle_fp_poly_eval(const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out)
{
    FP xx; memcpy(xx.mont, x, 24);
    const std::vector<FP> c = load<FP>(coeffs, n);
    FP r = FP::zero();
    for (size_t i = n; i-- > 0; ) { r *= xx; r += c[i]; }
    memcpy(out, r.mont, 24);
}
void oracle_fp_pow(const uint64_t *a, uint64_t e, uint64_t *out)
{
    FP x; memcpy(x.mont, a, 24);
    const FP r = x.pow(e);
    memcpy(out, r.mont, 24);
}

// row check (rowcheck.tcc:16-88)
int oracle_rowcheck_additive(int words, const uint64_t *az, const uint64_t *bz, const uint64_t *cz, const uint64_t *basis, size_t m,
                             const uint64_t *shift, size_t h, const uint64_t *constraint_shift, uint64_t *out)
{
    DISPATCH(words, {
        F cs; memcpy((void *)&cs, constraint_shift, sizeof(F));
        const size_t n = (size_t)1 << m;
        store<F>(out, rowcheck_additive<F>(load<F>(az, n), load<F>(bz, n), load<F>(cz, n), load_domain<F>(basis, m, shift), h, cs));
    });
    return 0;
}
void oracle_rowcheck_fp(const uint64_t *az, const uint64_t *bz, const uint64_t *cz, size_t order, const uint64_t *shift, size_t order_h,
                        const uint64_t *constraint_shift, uint64_t *out)
{
    FP cs; memcpy(cs.mont, constraint_shift, 24);
    store<FP>(out, rowcheck_multiplicative<FP>(load<FP>(az, order), load<FP>(bz, order), load<FP>(cz, order), load_coset(order, shift), order_h, cs));
}

// fz virtual oracle (r1cs_rs_iop.tcc:181-222)
int oracle_fz_additive(int words, const uint64_t *fw, const uint64_t *f1v, const uint64_t *basis, size_t m, const uint64_t *shift,
                       const uint64_t *ibasis, size_t idim, const uint64_t *ishift, uint64_t *out)
{
    DISPATCH(words, {
        const size_t n = (size_t)1 << m;
        store<F>(out, fz_additive<F>(load<F>(fw, n), load<F>(f1v, n), load_domain<F>(basis, m, shift), load_domain<F>(ibasis, idim, ishift)));
    });
    return 0;
}
void oracle_fz_fp(const uint64_t *fw, const uint64_t *f1v, size_t order, const uint64_t *shift, size_t input_order, const uint64_t *input_shift, uint64_t *out)
{
    FP is; memcpy(is.mont, input_shift, 24);
    store<FP>(out, fz_multiplicative<FP>(load<FP>(fw, order), load<FP>(f1v, order), load_coset(order, shift), input_order, is));
}

// sumcheck g oracle (sumcheck.tcc:58-119)
int oracle_sumcheck_g_additive(int words, const uint64_t *f, const uint64_t *h, const uint64_t *basis, size_t m, const uint64_t *shift,
                               const uint64_t *sbasis, size_t sdim, const uint64_t *sshift, const uint64_t *mu, uint64_t *out)
{
    DISPATCH(words, {
        F cs; memcpy((void *)&cs, mu, sizeof(F));
        const size_t n = (size_t)1 << m;
        store<F>(out, sumcheck_g_additive<F>(load<F>(f, n), load<F>(h, n), load_domain<F>(basis, m, shift), load_domain<F>(sbasis, sdim, sshift), cs));
    });
    return 0;
}
void oracle_sumcheck_g_fp(const uint64_t *f, const uint64_t *h, size_t order, const uint64_t *shift, size_t order_h, const uint64_t *sshift,
                          const uint64_t *mu, uint64_t *out)
{
    FP ss, m; memcpy(ss.mont, sshift, 24); memcpy(m.mont, mu, 24);
    store<FP>(out, sumcheck_g_multiplicative<FP>(load<FP>(f, order), load<FP>(h, order), load_coset(order, shift), order_h, ss, m));
}

// lincheck virtual oracle (basic_lincheck_aux.tcc:102-144); words == 0 selects the 181-bit prime field
int oracle_lincheck_combine(int words, const uint64_t *fz, const uint64_t *const *mz, size_t num, const uint64_t *r, const uint64_t *p1,
                            const uint64_t *p2, size_t n, uint64_t *out)
{
    if (words == 0) {
        std::vector<std::vector<FP>> M;
        for (size_t m = 0; m < num; ++m) M.push_back(load<FP>(mz[m], n));
        store<FP>(out, lincheck_combine<FP>(load<FP>(fz, n), M, load<FP>(r, num), load<FP>(p1, n), load<FP>(p2, n)));
        return 0;
    }
    DISPATCH(words, {
        std::vector<std::vector<F>> M;
        for (size_t m = 0; m < num; ++m) M.push_back(load<F>(mz[m], n));
        store<F>(out, lincheck_combine<F>(load<F>(fz, n), M, load<F>(r, num), load<F>(p1, n), load<F>(p2, n)));
    });
    return 0;
}


// ---- Aurora SNARK, prover and verifier (aurora.hpp) ----------------------------------------------------------------
// field: 1 = gf64, 3 = gf192 (additive arm), 0 = the 181-bit prime field edwards_Fr (multiplicative arm).
// The instance is generate_r1cs_example(2^log_constraints, num_inputs, 2^log_constraints - 1) from `seed`, as
// profiling/instrument_aurora_snark.cpp:108-110 calls it.
} // extern "C"

namespace {
std::vector<uint8_t> g_last_transcript;

#define AURORA_DISPATCH(field, CALL)                  \
    switch (field) {                                  \
    case 0: { typedef edwards_Fr F; CALL; } break;    \
    case 1: { typedef gf64 F; CALL; } break;          \
    case 3: { typedef gf192 F; CALL; } break;         \
    default: return -1;                               \
    }

template<typename F>
long aurora_prove_impl(size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization)
{
    const size_t n = (size_t)1 << log_constraints;
    const r1cs_example<F> ex = generate_r1cs_example<F>(n, num_inputs, n - 1, seed);
    const aurora_parameters<F> params(security, rs_extra, localization, n, n - 1, num_inputs);
    g_last_transcript = aurora_snark_prover<F>(ex.cs, ex.primary_input, ex.auxiliary_input, params).serialize();
    return (long)g_last_transcript.size();
}
template<typename F>
int aurora_verify_impl(size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization,
                       const uint8_t *bytes, size_t len, const uint64_t *primary_override)
{
    const size_t n = (size_t)1 << log_constraints;
    r1cs_example<F> ex = generate_r1cs_example<F>(n, num_inputs, n - 1, seed);
    if (primary_override) memcpy((void *)ex.primary_input.data(), primary_override, num_inputs * sizeof(F));
    const aurora_parameters<F> params(security, rs_extra, localization, n, n - 1, num_inputs);
    bcs_transcript<F> t;
    try { t = bcs_transcript<F>::deserialize(bytes, len); } catch (const std::exception &) { return 0; }
    return aurora_snark_verifier<F>(ex.cs, ex.primary_input, t, params) ? 1 : 0;
}
template<typename F>
int aurora_params_impl(size_t log_constraints, size_t num_inputs, size_t security, size_t rs_extra, size_t localization, uint64_t *out, size_t cap)
{
    const size_t n = (size_t)1 << log_constraints;
    const aurora_parameters<F> p(security, rs_extra, localization, n, n - 1, num_inputs);
    std::vector<uint64_t> v = { p.codeword_domain_dim, p.pow_bits, p.query_soundness_error_bits, p.interactive_soundness_error_bits,
                                p.max_tested_degree_bound, p.max_constraint_degree_bound, p.absolute_proximity_parameter, p.multi_lincheck_repetitions,
                                p.num_output_LDT_instances, p.fri_interactive_repetitions, p.fri_query_repetitions, p.localization_parameters.size() };
    for (size_t l : p.localization_parameters) v.push_back(l);
    if (v.size() > cap) return -1;
    memcpy(out, v.data(), v.size() * 8);
    return (int)v.size();
}
// ---- Fractal SNARK (fractal.hpp): indexer + prover; the index's Merkle roots are kept for oracle_fractal_index_roots ----
std::vector<uint8_t> g_last_index_roots;
template<typename F>
long fractal_prove_impl(size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization)
{
    const size_t n = (size_t)1 << log_constraints;
    const r1cs_example<F> ex = generate_r1cs_example<F>(n, num_inputs, n - 1, seed);
    const fractal_parameters<F> params(security, rs_extra, localization, ex.cs);
    const fractal_index<F> index = fractal_snark_indexer<F>(ex.cs, params);
    g_last_index_roots.clear();
    for (auto &r : index.MT_roots) g_last_index_roots.insert(g_last_index_roots.end(), r.begin(), r.end());
    g_last_transcript = fractal_snark_prover<F>(index, ex.cs, ex.primary_input, ex.auxiliary_input, params).serialize();
    return (long)g_last_transcript.size();
}
template<typename F>
int fractal_verify_impl(size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization,
                        const uint8_t *bytes, size_t len, const uint8_t *roots, size_t num_roots, const uint64_t *primary_override)
{
    const size_t n = (size_t)1 << log_constraints;
    r1cs_example<F> ex = generate_r1cs_example<F>(n, num_inputs, n - 1, seed);
    if (primary_override) memcpy((void *)ex.primary_input.data(), primary_override, num_inputs * sizeof(F));
    const fractal_parameters<F> params(security, rs_extra, localization, ex.cs);
    std::vector<digest_t> index_roots;
    for (size_t i = 0; i < num_roots; ++i) index_roots.push_back(digest_t(roots + i * DIGEST_LEN, roots + (i + 1) * DIGEST_LEN));
    bcs_transcript<F> t;
    try { t = bcs_transcript<F>::deserialize(bytes, len, num_roots); } catch (const std::exception &) { return 0; }
    return fractal_snark_verifier<F>(index_roots, ex.cs, ex.primary_input, t, params) ? 1 : 0;
}
template<typename F>
int fractal_params_impl(size_t log_constraints, size_t num_inputs, size_t security, size_t rs_extra, size_t localization, uint64_t *out, size_t cap)
{
    const size_t n = (size_t)1 << log_constraints;
    r1cs_system<F> shape;                           // the parameters read sizes and non-zero counts only: one entry per row, as the example has
    shape.num_inputs = num_inputs; shape.num_variables = n - 1;
    shape.A.assign(n, typename r1cs_system<F>::row(1, { 0, F::one() }));
    shape.B = shape.A; shape.C = shape.A;
    const fractal_parameters<F> p(security, rs_extra, localization, shape);
    std::vector<uint64_t> v = { p.codeword_domain_dim, p.pow_bits, p.query_soundness_error_bits, p.interactive_soundness_error_bits,
                                p.max_LDT_tested_degree_bound, p.max_constraint_degree_bound, p.absolute_proximity_parameter, p.holographic_lincheck_repetitions_,
                                p.num_output_LDT_instances, p.fri_interactive_repetitions, p.fri_query_repetitions, p.index_domain_dim, p.matrix_domain_dim,
                                p.localization_parameters.size() };
    for (size_t l : p.localization_parameters) v.push_back(l);
    if (v.size() > cap) return -1;
    memcpy(out, v.data(), v.size() * 8);
    return (int)v.size();
}
// one index oracle (matrix 0..2, which 0..3 = row, col, val, row*col) over the codeword domain; the twelve are computed once per
// argument tuple and kept
template<typename F>
int fractal_index_oracle_impl(size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization,
                              size_t matrix, size_t which, uint64_t *out)
{
    static std::vector<size_t> cached_key;
    static std::vector<std::vector<F>> cached;
    const std::vector<size_t> key = { log_constraints, num_inputs, (size_t)seed, security, rs_extra, localization };
    if (key != cached_key) {
        const size_t n = (size_t)1 << log_constraints;
        const r1cs_example<F> ex = generate_r1cs_example<F>(n, num_inputs, n - 1, seed);
        const fractal_parameters<F> params(security, rs_extra, localization, ex.cs);
        bcs_protocol<F> IOP(params.pow_bits);
        fractal_iop<F> full_protocol(IOP, ex.cs, params);
        cached = full_protocol.compute_index_oracles();
        cached_key = key;
    }
    const std::vector<F> &v = cached.at(4 * matrix + which);
    memcpy(out, (const void *)v.data(), v.size() * sizeof(F));
    return 0;
}
// the instance itself, so tests can compare the product's generator: z = (primary, auxiliary) then the C coefficients
template<typename F>
int r1cs_example_impl(size_t log_constraints, size_t num_inputs, uint64_t seed, uint64_t *z_out, uint64_t *c_index_out, uint64_t *c_coeff_out)
{
    const size_t n = (size_t)1 << log_constraints;
    const r1cs_example<F> ex = generate_r1cs_example<F>(n, num_inputs, n - 1, seed);
    std::vector<F> z(ex.primary_input);
    z.insert(z.end(), ex.auxiliary_input.begin(), ex.auxiliary_input.end());
    memcpy(z_out, (const void *)z.data(), z.size() * sizeof(F));
    for (size_t i = 0; i < n; ++i) {
        c_index_out[i] = ex.cs.C[i][0].first;
        memcpy((uint8_t *)c_coeff_out + i * sizeof(F), (const void *)&ex.cs.C[i][0].second, sizeof(F));
    }
    return 0;
}
} // namespace

extern "C" {

long oracle_aurora_prove(int field, size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization)
{
    try { AURORA_DISPATCH(field, return aurora_prove_impl<F>(log_constraints, num_inputs, seed, security, rs_extra, localization)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_aurora_prove: %s\n", e.what()); return -2; }
    return -1;
}
void oracle_aurora_fetch(uint8_t *dst) { if (!g_last_transcript.empty()) memcpy(dst, g_last_transcript.data(), g_last_transcript.size()); }

// primary_override: NULL, or num_inputs elements replacing the instance's primary input (a wrong statement must be rejected)
int oracle_aurora_verify(int field, size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization,
                         const uint8_t *transcript, size_t len, const uint64_t *primary_override)
{
    try { AURORA_DISPATCH(field, return aurora_verify_impl<F>(log_constraints, num_inputs, seed, security, rs_extra, localization, transcript, len, primary_override)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_aurora_verify: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_aurora_params(int field, size_t log_constraints, size_t num_inputs, size_t security, size_t rs_extra, size_t localization, uint64_t *out, size_t cap)
{
    try { AURORA_DISPATCH(field, return aurora_params_impl<F>(log_constraints, num_inputs, security, rs_extra, localization, out, cap)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_aurora_params: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_r1cs_example(int field, size_t log_constraints, size_t num_inputs, uint64_t seed, uint64_t *z_out, uint64_t *c_index_out, uint64_t *c_coeff_out)
{
    try { AURORA_DISPATCH(field, return r1cs_example_impl<F>(log_constraints, num_inputs, seed, z_out, c_index_out, c_coeff_out)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_r1cs_example: %s\n", e.what()); return -2; }
    return -1;
}


long oracle_fractal_prove(int field, size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization)
{
    try { AURORA_DISPATCH(field, return fractal_prove_impl<F>(log_constraints, num_inputs, seed, security, rs_extra, localization)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fractal_prove: %s\n", e.what()); return -2; }
    return -1;
}
// the index Merkle roots of the last oracle_fractal_prove: returns their number, copies 32 bytes each when dst is not null
long oracle_fractal_index_roots(uint8_t *dst)
{
    if (dst && !g_last_index_roots.empty()) memcpy(dst, g_last_index_roots.data(), g_last_index_roots.size());
    return (long)(g_last_index_roots.size() / DIGEST_LEN);
}
int oracle_fractal_verify(int field, size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization,
                          const uint8_t *transcript, size_t len, const uint8_t *roots, size_t num_roots, const uint64_t *primary_override)
{
    try { AURORA_DISPATCH(field, return fractal_verify_impl<F>(log_constraints, num_inputs, seed, security, rs_extra, localization, transcript, len, roots, num_roots, primary_override)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fractal_verify: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_fractal_params(int field, size_t log_constraints, size_t num_inputs, size_t security, size_t rs_extra, size_t localization, uint64_t *out, size_t cap)
{
    try { AURORA_DISPATCH(field, return fractal_params_impl<F>(log_constraints, num_inputs, security, rs_extra, localization, out, cap)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fractal_params: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_fractal_index_oracle(int field, size_t log_constraints, size_t num_inputs, uint64_t seed, size_t security, size_t rs_extra, size_t localization,
                                size_t matrix, size_t which, uint64_t *out)
{
    try { AURORA_DISPATCH(field, return fractal_index_oracle_impl<F>(log_constraints, num_inputs, seed, security, rs_extra, localization, matrix, which, out)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fractal_index_oracle: %s\n", e.what()); return -2; }
    return -1;
}

} // extern "C"

// ---- general instances: the constraint system as CSR triples (what a caller builds through r1cs_constraint_system::add_constraint,
// relations/r1cs.tcc:151-160: a row is the linear combination's term LIST — repeated indices and index 0, the constant 1, allowed,
// relations/variable.tcc:196-230) and the full variable assignment (primary inputs first).  Entry t of row i of matrix q is
// (col[q][t], coeff[q][t * words]) for row_ptr[q][i] <= t < row_ptr[q][i + 1]. ----
namespace {
struct csr_view {
    size_t num_constraints, num_variables, num_inputs;
    const uint64_t *const *row_ptr;
    const uint32_t *const *col;
    const uint64_t *const *coeff;
};
template<typename F>
r1cs_system<F> system_from_csr(const csr_view &v)
{
    r1cs_system<F> cs;
    cs.num_inputs = v.num_inputs;
    cs.num_variables = v.num_variables;
    std::vector<typename r1cs_system<F>::row> *M[3] = { &cs.A, &cs.B, &cs.C };
    for (int q = 0; q < 3; ++q) {
        M[q]->resize(v.num_constraints);
        for (size_t i = 0; i < v.num_constraints; ++i)
            for (uint64_t t = v.row_ptr[q][i]; t < v.row_ptr[q][i + 1]; ++t) {
                if (v.col[q][t] > v.num_variables) throw std::invalid_argument("column index exceeds the number of variables");
                F c;
                memcpy((void *)&c, (const uint8_t *)v.coeff[q] + t * sizeof(F), sizeof(F));
                (*M[q])[i].push_back({ (size_t)v.col[q][t], c });
            }
    }
    return cs;
}
template<typename F>
std::vector<F> load_elements(const uint64_t *src, size_t count)
{
    std::vector<F> out(count);
    if (count) memcpy((void *)out.data(), src, count * sizeof(F));
    return out;
}
template<typename F>
long aurora_prove_csr_impl(const csr_view &v, const uint64_t *assignment, size_t security, size_t rs_extra, size_t localization)
{
    const r1cs_system<F> cs = system_from_csr<F>(v);
    const std::vector<F> z = load_elements<F>(assignment, v.num_variables);
    const std::vector<F> primary(z.begin(), z.begin() + v.num_inputs), auxiliary(z.begin() + v.num_inputs, z.end());
    const aurora_parameters<F> params(security, rs_extra, localization, v.num_constraints, v.num_variables, v.num_inputs);
    g_last_transcript = aurora_snark_prover<F>(cs, primary, auxiliary, params).serialize();
    return (long)g_last_transcript.size();
}
template<typename F>
int aurora_verify_csr_impl(const csr_view &v, const uint64_t *primary_input, size_t security, size_t rs_extra, size_t localization, const uint8_t *bytes, size_t len)
{
    const r1cs_system<F> cs = system_from_csr<F>(v);
    const std::vector<F> primary = load_elements<F>(primary_input, v.num_inputs);
    const aurora_parameters<F> params(security, rs_extra, localization, v.num_constraints, v.num_variables, v.num_inputs);
    bcs_transcript<F> t;
    try { t = bcs_transcript<F>::deserialize(bytes, len); } catch (const std::exception &) { return 0; }
    return aurora_snark_verifier<F>(cs, primary, t, params) ? 1 : 0;
}
template<typename F>
long fractal_prove_csr_impl(const csr_view &v, const uint64_t *assignment, size_t security, size_t rs_extra, size_t localization)
{
    const r1cs_system<F> cs = system_from_csr<F>(v);
    const std::vector<F> z = load_elements<F>(assignment, v.num_variables);
    const std::vector<F> primary(z.begin(), z.begin() + v.num_inputs), auxiliary(z.begin() + v.num_inputs, z.end());
    const fractal_parameters<F> params(security, rs_extra, localization, cs);
    const fractal_index<F> index = fractal_snark_indexer<F>(cs, params);
    g_last_index_roots.clear();
    for (auto &r : index.MT_roots) g_last_index_roots.insert(g_last_index_roots.end(), r.begin(), r.end());
    g_last_transcript = fractal_snark_prover<F>(index, cs, primary, auxiliary, params).serialize();
    return (long)g_last_transcript.size();
}
template<typename F>
int fractal_verify_csr_impl(const csr_view &v, const uint64_t *primary_input, size_t security, size_t rs_extra, size_t localization, const uint8_t *bytes, size_t len,
                            const uint8_t *roots, size_t num_roots)
{
    const r1cs_system<F> cs = system_from_csr<F>(v);
    const std::vector<F> primary = load_elements<F>(primary_input, v.num_inputs);
    const fractal_parameters<F> params(security, rs_extra, localization, cs);
    std::vector<digest_t> index_roots;
    for (size_t i = 0; i < num_roots; ++i) index_roots.push_back(digest_t(roots + i * DIGEST_LEN, roots + (i + 1) * DIGEST_LEN));
    bcs_transcript<F> t;
    try { t = bcs_transcript<F>::deserialize(bytes, len, num_roots); } catch (const std::exception &) { return 0; }
    return fractal_snark_verifier<F>(index_roots, cs, primary, t, params) ? 1 : 0;
}
// r1cs_constraint_system::is_satisfied (relations/r1cs.tcc:112-147): <A_i, z> * <B_i, z> = <C_i, z> for every row, z = (1, assignment).
// Returns the number of violated rows; Mz_out (nullable): Az, Bz, Cz one after the other (create_Az_Bz_Cz_from_variable_assignment, :236-268).
template<typename F>
long r1cs_check_csr_impl(const csr_view &v, const uint64_t *assignment, uint64_t *Mz_out)
{
    const r1cs_system<F> cs = system_from_csr<F>(v);
    std::vector<F> z(1, F::one());
    const std::vector<F> rest = load_elements<F>(assignment, v.num_variables);
    z.insert(z.end(), rest.begin(), rest.end());
    const std::vector<F> Az = sparse_times_vector<F>(cs.A, z), Bz = sparse_times_vector<F>(cs.B, z), Cz = sparse_times_vector<F>(cs.C, z);
    long violated = 0;
    for (size_t i = 0; i < v.num_constraints; ++i) if (!(Az[i] * Bz[i] == Cz[i])) ++violated;
    if (Mz_out) {
        const std::vector<F> *three[3] = { &Az, &Bz, &Cz };
        for (int q = 0; q < 3; ++q) memcpy((uint8_t *)Mz_out + q * v.num_constraints * sizeof(F), (const void *)three[q]->data(), v.num_constraints * sizeof(F));
    }
    return violated;
}
} // namespace

extern "C" {

#define CSR_VIEW const csr_view v = { num_constraints, num_variables, num_inputs, row_ptr, col, coeff }
long oracle_aurora_prove_csr(int field, size_t num_constraints, size_t num_variables, size_t num_inputs, const uint64_t *const *row_ptr, const uint32_t *const *col,
                             const uint64_t *const *coeff, const uint64_t *assignment, size_t security, size_t rs_extra, size_t localization)
{
    CSR_VIEW;
    try { AURORA_DISPATCH(field, return aurora_prove_csr_impl<F>(v, assignment, security, rs_extra, localization)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_aurora_prove_csr: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_aurora_verify_csr(int field, size_t num_constraints, size_t num_variables, size_t num_inputs, const uint64_t *const *row_ptr, const uint32_t *const *col,
                             const uint64_t *const *coeff, const uint64_t *primary_input, size_t security, size_t rs_extra, size_t localization,
                             const uint8_t *transcript, size_t len)
{
    CSR_VIEW;
    try { AURORA_DISPATCH(field, return aurora_verify_csr_impl<F>(v, primary_input, security, rs_extra, localization, transcript, len)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_aurora_verify_csr: %s\n", e.what()); return -2; }
    return -1;
}
long oracle_fractal_prove_csr(int field, size_t num_constraints, size_t num_variables, size_t num_inputs, const uint64_t *const *row_ptr, const uint32_t *const *col,
                              const uint64_t *const *coeff, const uint64_t *assignment, size_t security, size_t rs_extra, size_t localization)
{
    CSR_VIEW;
    try { AURORA_DISPATCH(field, return fractal_prove_csr_impl<F>(v, assignment, security, rs_extra, localization)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fractal_prove_csr: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_fractal_verify_csr(int field, size_t num_constraints, size_t num_variables, size_t num_inputs, const uint64_t *const *row_ptr, const uint32_t *const *col,
                              const uint64_t *const *coeff, const uint64_t *primary_input, size_t security, size_t rs_extra, size_t localization,
                              const uint8_t *transcript, size_t len, const uint8_t *roots, size_t num_roots)
{
    CSR_VIEW;
    try { AURORA_DISPATCH(field, return fractal_verify_csr_impl<F>(v, primary_input, security, rs_extra, localization, transcript, len, roots, num_roots)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fractal_verify_csr: %s\n", e.what()); return -2; }
    return -1;
}
long oracle_r1cs_check_csr(int field, size_t num_constraints, size_t num_variables, size_t num_inputs, const uint64_t *const *row_ptr, const uint32_t *const *col,
                           const uint64_t *const *coeff, const uint64_t *assignment, uint64_t *Mz_out)
{
    CSR_VIEW;
    try { AURORA_DISPATCH(field, return r1cs_check_csr_impl<F>(v, assignment, Mz_out)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_r1cs_check_csr: %s\n", e.what()); return -2; }
    return -1;
}
#undef CSR_VIEW

// the block timers of field.hpp: "name\tseconds\tcalls\n" per block into buf (returns the length needed); reset != 0 clears them afterwards
size_t oracle_block_times(char *buf, size_t cap, int reset)
{
    std::string out;
    for (auto &kv : block_times::table()) {
        char line[256];
        snprintf(line, sizeof line, "%s\t%.9f\t%zu\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf && cap) { const size_t n = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), n); buf[n] = 0; }
    if (reset) block_times::table().clear();
    return out.size() + 1;
}

// FRI-only SNARK (aurora.hpp FRI_snark_*): the polynomial's coefficients are seeded_element(seed, i), i < 2^(dim - rs_extra)
} // extern "C"

namespace {
template<typename F>
FRI_snark_parameters<F> make_fri_params(size_t dim, size_t rs_extra, size_t localization, size_t interactions, size_t queries)
{
    FRI_snark_parameters<F> p;
    p.codeword_domain_dim = dim; p.RS_extra_dimensions = rs_extra; p.num_interactive_repetitions = interactions; p.num_query_repetitions = queries;
    p.localization_parameters = localization_parameter_to_array(localization, dim, rs_extra);
    return p;
}
template<typename F>
long fri_prove_impl(size_t dim, size_t rs_extra, size_t localization, size_t interactions, size_t queries, uint64_t seed)
{
    const FRI_snark_parameters<F> p = make_fri_params<F>(dim, rs_extra, localization, interactions, queries);
    std::vector<F> coeffs(p.poly_degree_bound());
    for (size_t i = 0; i < coeffs.size(); ++i) coeffs[i] = seeded_element(seed, i, (const F *)nullptr);
    g_last_transcript = FRI_snark_prover<F>(coeffs, p).serialize();
    return (long)g_last_transcript.size();
}
template<typename F>
int fri_verify_impl(size_t dim, size_t rs_extra, size_t localization, size_t interactions, size_t queries, const uint8_t *bytes, size_t len)
{
    const FRI_snark_parameters<F> p = make_fri_params<F>(dim, rs_extra, localization, interactions, queries);
    bcs_transcript<F> t;
    try { t = bcs_transcript<F>::deserialize(bytes, len); } catch (const std::exception &) { return 0; }
    return FRI_snark_verifier<F>(t, p) ? 1 : 0;
}
} // namespace

extern "C" {

long oracle_fri_snark_prove(int field, size_t dim, size_t rs_extra, size_t localization, size_t interactions, size_t queries, uint64_t seed)
{
    try { AURORA_DISPATCH(field, return fri_prove_impl<F>(dim, rs_extra, localization, interactions, queries, seed)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fri_snark_prove: %s\n", e.what()); return -2; }
    return -1;
}
int oracle_fri_snark_verify(int field, size_t dim, size_t rs_extra, size_t localization, size_t interactions, size_t queries,
                            const uint8_t *transcript, size_t len)
{
    try { AURORA_DISPATCH(field, return fri_verify_impl<F>(dim, rs_extra, localization, interactions, queries, transcript, len)); }
    catch (const std::exception &e) { fprintf(stderr, "oracle_fri_snark_verify: %s\n", e.what()); return -2; }
    return -1;
}

} // extern "C"
