// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's proof-of-work grind, libiop/bcs/pow.tcc (citations relative to /root/reference).
// The reference's tests hold no known answers for it (tests/snark/test_pow.cpp:13-56 checks that solve_pow's answer
// verifies); what pins this restatement is the hash underneath (BLAKE2b: RFC 7693 vectors; Poseidon:
// test_poseidon.cpp's known answers) plus the literal candidate order below.
#pragma once
#include <cstdint>
#include <cstring>
#include "merkle.hpp"
#include "poseidon.hpp"

namespace oracle {

// pow_parameters::pow_bitlen (pow.tcc:21-32): work_parameter - floor(log2(cost_per_hash)); libff::log2 is the ceiling
static inline size_t pow_bitlen(size_t work_parameter, size_t cost_per_hash)
{
    size_t log_cost = 0;
    while (((size_t)1 << log_cost) < cost_per_hash) ++log_cost;
    if (((size_t)1 << log_cost) > cost_per_hash) log_cost -= 1;
    return work_parameter - log_cost;
}

// verify_pow + verify_pow_internal, binary digests (pow.tcc:111-119,143-162): the last 8-byte word of
// H(challenge || pow), low pow_bitlen bits, must be <= pow_upperbound() = 0
static inline bool pow_verify_blake2b(const uint8_t *challenge, const uint8_t *pow, size_t bitlen)
{
    uint8_t h[DIGEST_LEN];
    two_to_one(challenge, pow, h);
    uint64_t w;
    memcpy(&w, h + DIGEST_LEN - 8, 8);
    return (w & (uint64_t)((1 << bitlen) - 1)) == 0;
}

// solve_pow_internal, binary digests (pow.tcc:86-103): first candidate is the challenge itself, then the challenge with
// its last word overwritten by 0, 1, 2, ...; returns the number of verify calls made
static inline uint64_t pow_solve_blake2b(const uint8_t *challenge, size_t bitlen, uint8_t *pow)
{
    memcpy(pow, challenge, DIGEST_LEN);
    uint64_t pow_int = 0, calls = 1;
    while (!pow_verify_blake2b(challenge, pow, bitlen)) {
        memcpy(pow + DIGEST_LEN - 8, &pow_int, 8);
        pow_int += 1;
        calls += 1;
    }
    return calls;
}

// algebraic digests (pow.tcc:73-84,129-141): pow counts up from FieldT::zero(); word 0 of the hash's canonical
// integer (libff::get_word_of_field_elem(hash, 0) = as_bigint().data[0]; libff is absent here: recalled fact)
template<typename F>
bool pow_verify_poseidon(const poseidon_params<F> &P, const F &challenge, const F &pow, size_t bitlen)
{
    const F h = poseidon_two_to_one<F>(P, challenge, pow);
    uint64_t c[F::N];
    h.to_canonical(c);
    return (c[0] & (uint64_t)((1 << bitlen) - 1)) == 0;
}

template<typename F>
uint64_t pow_solve_poseidon(const poseidon_params<F> &P, const F &challenge, size_t bitlen, F &pow)
{
    pow = F::zero();
    uint64_t calls = 1;
    while (!pow_verify_poseidon<F>(P, challenge, pow, bitlen)) {
        pow += F::one();
        calls += 1;
    }
    return calls;
}

} // namespace oracle
