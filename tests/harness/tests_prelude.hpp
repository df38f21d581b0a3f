// TEST INFRASTRUCTURE ONLY (tests/harness): force-included ahead of one of libiop's own test files in the stubbed build — libiop's headers (shadowed by the
// declarations of INTEGRATION.md's stubs), then the stub definitions, so that the test file's text itself stays untouched.
#pragma once
#include <libff/algebra/fields/binary/gf192.hpp>
#include <libff/algebra/curves/edwards/edwards_pp.hpp>
#include "libiop/algebra/fft.hpp"
#include "libiop/protocols/ldt/fri/fri_aux.hpp"
#include "libiop/protocols/ldt/ldt_reducer_aux.hpp"
#include "libiop/bcs/merkle_tree.hpp"
#include "libiop/bcs/pow.hpp"
#include "libiop/protocols/encoded/common/rowcheck.hpp"
#include "libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.hpp"
#include "libiop/protocols/encoded/sumcheck/sumcheck.hpp"
#include "libiop/protocols/encoded/lincheck/basic_lincheck_aux.hpp"
#include "stubs.inc"
