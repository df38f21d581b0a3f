// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// BLAKE2b as specified in RFC 7693, written from the RFC (the reference calls libsodium's
// crypto_generichash_blake2b, libiop/bcs/hashing/blake2b.tcc:140-160, blake2b.cpp:28-48; libsodium is
// not assumed on the GPU box).  Pinned in tests/test_oracle_blake2b.py against the RFC 7693
// appendix-A vector and against Python's hashlib.blake2b (unkeyed and keyed, many lengths).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace oracle {

static const uint64_t blake2b_iv[8] = {
    0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
    0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };

static const uint8_t blake2b_sigma[12][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 },
    { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 },
    { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 },
    { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 },
    { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 },
    { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 },
    { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 } };

static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static inline void blake2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t_lo, uint64_t t_hi, bool last)
{
    uint64_t m[16], v[16];
    memcpy(m, block, 128);                        // little-endian host
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = blake2b_iv[i]; }
    v[12] ^= t_lo; v[13] ^= t_hi;
    if (last) v[14] = ~v[14];
#define ORACLE_G(a, b, c, d, x, y) \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = blake2b_sigma[r];
        ORACLE_G(0, 4,  8, 12, m[s[0]],  m[s[1]]);
        ORACLE_G(1, 5,  9, 13, m[s[2]],  m[s[3]]);
        ORACLE_G(2, 6, 10, 14, m[s[4]],  m[s[5]]);
        ORACLE_G(3, 7, 11, 15, m[s[6]],  m[s[7]]);
        ORACLE_G(0, 5, 10, 15, m[s[8]],  m[s[9]]);
        ORACLE_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        ORACLE_G(2, 7,  8, 13, m[s[12]], m[s[13]]);
        ORACLE_G(3, 4,  9, 14, m[s[14]], m[s[15]]);
    }
#undef ORACLE_G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

// out[outlen] = BLAKE2b(in[inlen], key[keylen]); 1 <= outlen <= 64, keylen <= 64.
static inline void blake2b(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen,
                           const uint8_t *key = nullptr, size_t keylen = 0)
{
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = blake2b_iv[i];
    h[0] ^= 0x01010000ull ^ ((uint64_t)keylen << 8) ^ (uint64_t)outlen;

    uint8_t block[128];
    uint64_t t = 0;
    if (keylen > 0) {
        memset(block, 0, 128);
        memcpy(block, key, keylen);
        if (inlen == 0) { blake2b_compress(h, block, 128, 0, true); goto done; }
        t = 128;
        blake2b_compress(h, block, t, 0, false);
    }
    while (inlen > 128) {
        t += 128;
        blake2b_compress(h, in, t, 0, false);
        in += 128; inlen -= 128;
    }
    memset(block, 0, 128);
    if (inlen) memcpy(block, in, inlen);
    t += inlen;
    blake2b_compress(h, block, t, 0, true);
done:
    memcpy(out, h, outlen);
}

} // namespace oracle
