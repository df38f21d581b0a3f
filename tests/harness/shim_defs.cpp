// TEST INFRASTRUCTURE ONLY (tests/harness): the static members of the stand-in field classes, and a main() for libiop's test files (which rely on gtest_main).
#include <libff/algebra/fields/binary/gf64.hpp>
#include <libff/algebra/fields/binary/gf192.hpp>
#include <libff/algebra/curves/edwards/edwards_pp.hpp>
#include <libff/algebra/curves/alt_bn128/alt_bn128_pp.hpp>
#include <libff/common/profiling.hpp>

namespace libff {
gf64 gf64::multiplicative_generator = gf64(2);
gf192 gf192::multiplicative_generator = gf192(2);
bigint<3> edwards_Fr::mod;
edwards_Fr edwards_Fr::multiplicative_generator, edwards_Fr::root_of_unity;
bigint<4> alt_bn128_Fr::mod;
alt_bn128_Fr alt_bn128_Fr::multiplicative_generator, alt_bn128_Fr::root_of_unity;
bool inhibit_profiling_info = true, inhibit_profiling_counters = true;
}
