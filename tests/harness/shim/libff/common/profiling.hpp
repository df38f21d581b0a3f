// TEST INFRASTRUCTURE ONLY (tests/harness): no-op stand-in for libff/common/profiling.hpp.
#pragma once
#include <string>
#include <cstdio>
namespace libff {
inline void start_profiling() {}
inline void enter_block(const std::string &, bool = false) {}
inline void leave_block(const std::string &, bool = false) {}
inline void print_indent() {}
inline void print_separator() {}
inline void print_header(const char *) {}
inline long long get_nsec_time() { return 0; }
extern bool inhibit_profiling_info;
extern bool inhibit_profiling_counters;
} // namespace libff
