// Plain field-element types for C++ callers that do not link libff: 24 raw bytes with the layout of libff::gf192 (three
// little-endian words, polynomial basis) / libff::edwards_Fr (three Montgomery limbs).  They provide exactly what the mirror asks of
// a FieldT — construction from an integer, ==, + — through the library's host helpers; a libiop integration uses libff's own types
// instead.
#pragma once
#include "iop.hpp"

namespace libiop_amd {

struct gf192_element {
    uint64_t w[3];
    gf192_element() : w{ 0, 0, 0 } {}
    explicit gf192_element(uint64_t v) : w{ v, 0, 0 } {}
    bool operator==(const gf192_element &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2]; }
    bool operator!=(const gf192_element &o) const { return !(*this == o); }
    gf192_element operator+(const gf192_element &o) const { gf192_element r; for (int i = 0; i < 3; ++i) r.w[i] = w[i] ^ o.w[i]; return r; }
    gf192_element &operator+=(const gf192_element &o) { for (int i = 0; i < 3; ++i) w[i] ^= o.w[i]; return *this; }
};
template<> struct field_kind<gf192_element> { static const field_subset_type type = affine_subspace_type; };

struct edwards_Fr_element {
    uint64_t w[3];
    edwards_Fr_element() : w{ 0, 0, 0 } {}
    explicit edwards_Fr_element(uint64_t v) { check(iopx_fp3_from_uint(v, w)); }
    bool operator==(const edwards_Fr_element &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2]; }
    bool operator!=(const edwards_Fr_element &o) const { return !(*this == o); }
    edwards_Fr_element operator+(const edwards_Fr_element &o) const { edwards_Fr_element r; check(iopx_fp3_host_add(w, o.w, r.w)); return r; }
};
template<> struct field_kind<edwards_Fr_element> { static const field_subset_type type = multiplicative_coset_type; };

} // namespace libiop_amd
