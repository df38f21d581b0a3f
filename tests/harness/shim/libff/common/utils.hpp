// TEST INFRASTRUCTURE ONLY (tests/harness): a stand-in for libff/common/utils.hpp — libff is an empty submodule of the reference tree.
// Only what libiop's sources name; nothing here is part of the product and nothing here pins libff's behaviour.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <numeric>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <gmp.h>

namespace libff {

typedef std::vector<bool> bit_vector;

inline std::size_t log2(std::size_t n)              // ceil(log2 n), 0 for n <= 1
{
    std::size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) { n >>= 1; ++r; }
    return r;
}
inline std::size_t bitreverse(std::size_t n, const std::size_t l)
{
    std::size_t r = 0;
    for (std::size_t k = 0; k < l; ++k) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}
inline bool is_power_of_2(const std::size_t n) { return n != 0 && (n & (n - 1)) == 0; }
inline std::size_t round_to_next_power_of_2(const std::size_t n) { return (std::size_t)1 << log2(n); }
template<typename... Types> void UNUSED(Types &&...) {}
template<typename T> void print_vector(const std::vector<T> &vec) { for (auto &v : vec) std::cout << v << " "; std::cout << std::endl; }

} // namespace libff
