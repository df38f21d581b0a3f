// Host-side GF(2^192) arithmetic used only to prepare the per-domain constants (recursed bases,
// shifts, fold multipliers: O(m^2) field operations per plan).  Portable C++ (no intrinsics): the
// hot path runs on the GPU, this never touches codeword-sized data.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace iopx {

struct hgf192 {
    uint64_t w[3];

    hgf192() { w[0] = w[1] = w[2] = 0; }
    explicit hgf192(uint64_t v) { w[0] = v; w[1] = w[2] = 0; }
    static hgf192 from_words(const uint64_t *p) { hgf192 r; r.w[0] = p[0]; r.w[1] = p[1]; r.w[2] = p[2]; return r; }
    static hgf192 zero() { return hgf192(); }
    static hgf192 one() { return hgf192(1); }

    bool is_zero() const { return (w[0] | w[1] | w[2]) == 0; }
    bool operator==(const hgf192 &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2]; }
    bool operator!=(const hgf192 &o) const { return !(*this == o); }
    bool operator<(const hgf192 &o) const { return memcmp(w, o.w, sizeof(w)) < 0; }

    hgf192 operator+(const hgf192 &o) const { hgf192 r; for (int i = 0; i < 3; ++i) r.w[i] = w[i] ^ o.w[i]; return r; }
    hgf192 &operator+=(const hgf192 &o) { for (int i = 0; i < 3; ++i) w[i] ^= o.w[i]; return *this; }

    hgf192 operator*(const hgf192 &o) const
    {
        // 4-bit window comb over the 48 nibbles of o; table of (*this) * u for u < 16 (195 bits each)
        uint64_t tab[16][4];
        memset(tab, 0, sizeof(tab));
        for (int i = 0; i < 3; ++i) tab[1][i] = w[i];
        for (int u = 2; u < 16; u += 2) {
            for (int i = 3; i > 0; --i) tab[u][i] = (tab[u >> 1][i] << 1) | (tab[u >> 1][i - 1] >> 63);
            tab[u][0] = tab[u >> 1][0] << 1;
            for (int i = 0; i < 4; ++i) tab[u + 1][i] = tab[u][i] ^ tab[1][i];
        }
        uint64_t c[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int nib = 15; nib >= 0; --nib) {
            for (int i = 6; i > 0; --i) c[i] = (c[i] << 4) | (c[i - 1] >> 60);
            c[0] <<= 4;
            for (int k = 0; k < 3; ++k) {
                const uint64_t *t = tab[(o.w[k] >> (4 * nib)) & 15];
                for (int i = 0; i < 4; ++i) c[k + i] ^= t[i];
            }
        }
        // reduce modulo x^192 + x^7 + x^2 + x + 1 (c[6] is always zero: the product has 383 bits)
        for (int i = 5; i >= 3; --i) {
            const uint64_t t = c[i];
            c[i - 3] ^= t ^ (t << 1) ^ (t << 2) ^ (t << 7);
            c[i - 2] ^= (t >> 63) ^ (t >> 62) ^ (t >> 57);
        }
        hgf192 r; r.w[0] = c[0]; r.w[1] = c[1]; r.w[2] = c[2];
        return r;
    }
    hgf192 &operator*=(const hgf192 &o) { *this = *this * o; return *this; }

    // squaring is GF(2)-linear: spread the bits (bit i -> bit 2i) and reduce
    hgf192 squared() const
    {
        uint64_t c[6];
        for (int i = 0; i < 3; ++i) {
            uint64_t lo = w[i] & 0xffffffffull, hi = w[i] >> 32;
            lo = (lo | (lo << 16)) & 0x0000ffff0000ffffull; hi = (hi | (hi << 16)) & 0x0000ffff0000ffffull;
            lo = (lo | (lo << 8)) & 0x00ff00ff00ff00ffull;  hi = (hi | (hi << 8)) & 0x00ff00ff00ff00ffull;
            lo = (lo | (lo << 4)) & 0x0f0f0f0f0f0f0f0full;  hi = (hi | (hi << 4)) & 0x0f0f0f0f0f0f0f0full;
            lo = (lo | (lo << 2)) & 0x3333333333333333ull;  hi = (hi | (hi << 2)) & 0x3333333333333333ull;
            lo = (lo | (lo << 1)) & 0x5555555555555555ull;  hi = (hi | (hi << 1)) & 0x5555555555555555ull;
            c[2 * i] = lo; c[2 * i + 1] = hi;
        }
        for (int i = 5; i >= 3; --i) {
            const uint64_t t = c[i];
            c[i - 3] ^= t ^ (t << 1) ^ (t << 2) ^ (t << 7);
            c[i - 2] ^= (t >> 63) ^ (t >> 62) ^ (t >> 57);
        }
        hgf192 r; r.w[0] = c[0]; r.w[1] = c[1]; r.w[2] = c[2];
        return r;
    }

    // a^(2^192 - 2) = (a^(2^191 - 1))^2 by an Itoh–Tsujii addition chain on the exponent 191:
    // beta_k = a^(2^k - 1), beta_{2k} = beta_k^(2^k) * beta_k, beta_{k+1} = beta_k^2 * a
    hgf192 inverse() const
    {
        hgf192 beta = *this;
        int k = 1;
        for (int bit = 6; bit >= 0; --bit) {        // 191 = 0b10111111
            hgf192 t = beta;
            for (int i = 0; i < k; ++i) t = t.squared();
            beta = t * beta;
            k *= 2;
            if ((191 >> bit) & 1) { beta = beta.squared() * *this; k += 1; }
        }
        return beta.squared();
    }
};

// coefficient i >= 1 multiplies X^(2^(i-1)); slot 0 is the constant term
inline hgf192 linearized_eval(const std::vector<hgf192> &c, const hgf192 &x)
{
    hgf192 r = c.empty() ? hgf192::zero() : c[0];
    hgf192 xp = x;
    for (size_t i = 1; i < c.size(); ++i) { r += c[i] * xp; xp = xp.squared(); }
    return r;
}

// The subspace polynomial of span(basis[0..dim)): prod_{v in span} (X - v), a linearized polynomial (coeff[i] multiplies
// X^(2^i)), built factor by factor as Z <- Z(X) (Z(X) + Z(b)) (libiop/algebra/polynomials/vanishing_polynomial.tcc:373-395).
// The vanishing polynomial of the affine subspace span + shift is eval(X) + eval(shift).
struct SubspacePoly {
    std::vector<hgf192> coeff;
    SubspacePoly(const uint64_t *basis, size_t dim) : coeff(1, hgf192::one())
    {
        for (size_t k = 0; k < dim; ++k) {
            const hgf192 zb = eval(hgf192::from_words(basis + 3 * k));
            std::vector<hgf192> nxt(coeff.size() + 1, hgf192::zero());
            for (size_t i = 0; i < coeff.size(); ++i) { nxt[i + 1] += coeff[i].squared(); nxt[i] += coeff[i] * zb; }
            coeff.swap(nxt);
        }
    }
    hgf192 eval(const hgf192 &x) const
    {
        hgf192 r = hgf192::zero(), xp = x;
        for (size_t i = 0; i < coeff.size(); ++i) { r += coeff[i] * xp; xp = xp.squared(); }
        return r;
    }
};

// The subspace polynomial of a domain and its values at the domain's own vectors are the same in every proof over that domain, and building
// them (dim^2 host products, then dim products per value) sat on the prover's critical path between two rounds: kept per basis.
struct CachedSubspacePoly {
    SubspacePoly poly;
    std::map<std::array<uint64_t, 3>, hgf192> values;
    std::mutex mu;
    CachedSubspacePoly(const uint64_t *basis, size_t dim) : poly(basis, dim) {}
    hgf192 eval(const hgf192 &x)
    {
        const std::array<uint64_t, 3> key = { x.w[0], x.w[1], x.w[2] };
        std::lock_guard<std::mutex> lk(mu);
        auto it = values.find(key);
        if (it != values.end()) return it->second;
        if (values.size() >= 65536) values.clear();
        const hgf192 v = poly.eval(x);
        values.emplace(key, v);
        return v;
    }
};

// Returned by value: a caller keeps its entry alive while another thread's 65th distinct basis clears the cache.
inline std::shared_ptr<CachedSubspacePoly> cached_subspace_poly(const uint64_t *basis, size_t dim)
{
    static std::mutex mu;
    static std::map<std::vector<uint64_t>, std::shared_ptr<CachedSubspacePoly>> cache;
    std::vector<uint64_t> key(basis, basis + 3 * dim);
    key.push_back(dim);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 64) cache.clear();
        it = cache.emplace(key, std::make_shared<CachedSubspacePoly>(basis, dim)).first;
    }
    return it->second;
}


// Device layout of a subset-sum table (gf192_dev.h subset_sum_ext): the m + 1 entries, then 512 pre-summed entries for index bits
// 8..16 and 512 for bits 17..25.
inline void append_subset_table_with_ext(std::vector<uint64_t> &out, const std::vector<hgf192> &entries)
{
    const size_t m = entries.size() - 1;
    for (const hgf192 &e : entries) out.insert(out.end(), e.w, e.w + 3);
    for (int level = 0; level < 2; ++level) {
        const size_t first = 8 + 9 * (size_t)level;
        for (size_t q = 0; q < 512; ++q) {
            hgf192 acc = hgf192::zero();
            for (size_t k = 0; k < 9 && first + k < m; ++k) if ((q >> k) & 1) acc += entries[1 + first + k];
            out.insert(out.end(), acc.w, acc.w + 3);
        }
    }
}
static const size_t SUBSET_TABLE_WORDS_EXTRA = 3 * 1024;

} // namespace iopx
