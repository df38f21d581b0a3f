// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's Poseidon sponge and the algebraic leaf / two-to-one hashes built on it
// (libiop/bcs/hashing/poseidon.tcc:159-297, algebraic_sponge.tcc:18-100,110-125,220-265), over alt_bn128 Fr.
// Pinned by the reference's own known answers (libiop/tests/snark/test_poseidon.cpp:55,65,97; tests/golden/
// poseidon_kat.json).
#pragma once
#include <cassert>
#include <vector>
#include "fp.hpp"

namespace oracle {

template<typename F>
struct poseidon_params {
    size_t full_rounds, partial_rounds, alpha, rate, state_size;
    bool near_mds;
    std::vector<std::vector<F>> mds, ark;       // mds: t x t ; ark: (R_F + R_P) x t
};

template<typename F>
struct poseidon_sponge {
    poseidon_params<F> P;
    std::vector<F> state;
    size_t next_unsqueezed = 0;
    bool currently_absorbing = false;

    explicit poseidon_sponge(const poseidon_params<F> &p) : P(p), state(p.state_size, F::zero()) {}

    F raise_to_alpha(const F &x) const                 // poseidon.tcc:159-193
    {
        if (P.alpha == 17) { F t = x * x; t *= t; t *= t; t *= t; t *= x; return t; }
        if (P.alpha == 5) { F t = x * x; t *= t; return x * t; }
        if (P.alpha == 3) { F t = x * x; return x * t; }
        return x.pow(P.alpha);
    }
    void mix()                                          // poseidon.tcc:195-239
    {
        const size_t t = P.state_size;
        if (P.near_mds && t == 3) {
            const F x = state[0];
            state[0] += state[2];
            state[2] += state[1];
            state[1] += x;
        } else if (P.near_mds && t == 4) {
            const F sum = (state[0] + state[1]) + (state[2] + state[3]);
            for (size_t i = 0; i < 4; ++i) state[i] = sum - state[i];
        } else {
            std::vector<F> out(t, F::zero());
            for (size_t r = 0; r < t; ++r)
                for (size_t c = 0; c < t; ++c) out[r] += P.mds[r][c] * state[c];
            state = out;
        }
    }
    void full_round(size_t rid)                         // :241-254
    {
        for (size_t i = 0; i < P.state_size; ++i) { state[i] += P.ark[rid][i]; state[i] = raise_to_alpha(state[i]); }
        mix();
    }
    void partial_round(size_t rid)                      // :256-271
    {
        for (size_t i = 0; i < P.state_size; ++i) state[i] += P.ark[rid][i];
        state[P.state_size - 1] = raise_to_alpha(state[P.state_size - 1]);
        mix();
    }
    void permute()                                      // :273-297
    {
        size_t round = 0;
        for (size_t i = 0; i < P.full_rounds / 2; ++i) full_round(round++);
        for (size_t i = 0; i < P.partial_rounds; ++i) partial_round(round++);
        for (size_t i = 0; i < P.full_rounds / 2; ++i) full_round(round++);
    }
    void reset()                                        // :299-309
    {
        for (F &s : state) s = F::zero();
        next_unsqueezed = 0;
        currently_absorbing = false;
    }
    void absorb(const std::vector<F> &in)               // algebraic_sponge.tcc:18-62
    {
        if (currently_absorbing) permute();
        size_t begin = 0;
        while (in.size() - begin > P.rate) {
            for (size_t i = 0; i < P.rate; ++i) state[i] += in[i + begin];
            permute();
            begin += P.rate;
        }
        for (size_t i = 0; i < in.size() - begin; ++i) state[i] += in[i + begin];
        currently_absorbing = true;
    }
    std::vector<F> squeeze(size_t n)                    // algebraic_sponge.tcc:64-100
    {
        std::vector<F> out(n, F::zero());
        if (currently_absorbing) { next_unsqueezed = 0; currently_absorbing = false; }
        size_t idx = 0;
        while (true) {
            if (next_unsqueezed == 0) permute();
            while (next_unsqueezed < P.rate && idx < n) out[idx++] = state[next_unsqueezed++];
            if (idx == n) return out;
            next_unsqueezed = 0;
        }
    }
};

// algebraic_leafhash::hash (algebraic_sponge.tcc:220-228)
template<typename F>
F poseidon_leafhash(const poseidon_params<F> &p, const std::vector<F> &leaf)
{
    poseidon_sponge<F> sp(p);
    sp.absorb(leaf);
    return sp.squeeze(1)[0];
}

// algebraic_two_to_one_hash::hash (algebraic_sponge.tcc:256-265)
template<typename F>
F poseidon_two_to_one(const poseidon_params<F> &p, const F &l, const F &r)
{
    poseidon_sponge<F> sp(p);
    sp.state[0] = l;
    sp.state[1] = r;
    return sp.squeeze(1)[0];
}

// string_to_field_elem, multiplicative case (algebraic_sponge.tcc:110-125): 8-byte words of the salt, first word
// becomes the MOST significant limb, then FieldT(bigint)
template<typename F>
F poseidon_salt_to_field(const uint8_t *salt)
{
    uint64_t c[F::N];
    for (int i = 0; i < F::N; ++i) memcpy(&c[F::N - i - 1], salt + 8 * i, 8);
    return F::from_canonical(c);      // Montgomery reduction also reduces values >= p (as libff's FieldT(bigint) does)
}

} // namespace oracle
