// TEST INFRASTRUCTURE ONLY (tests/harness): libiop's r1cs_rs_iop.hpp includes this googletest header for FRIEND_TEST.
#pragma once
#define FRIEND_TEST(test_case_name, test_name) friend class test_case_name##_##test_name##_Test
