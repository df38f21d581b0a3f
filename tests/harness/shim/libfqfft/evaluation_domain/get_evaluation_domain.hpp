// TEST INFRASTRUCTURE ONLY (tests/harness): libiop includes this header and uses nothing of it beyond basic_radix2_domain.
#pragma once
#include <libfqfft/evaluation_domain/domains/basic_radix2_domain.hpp>
