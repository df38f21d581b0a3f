// TEST INFRASTRUCTURE ONLY (tests/harness): what gtest_main provides to libiop's test files, plus the kernel library's initialisation in the stubbed build.
#include <gtest/gtest.h>
#ifdef HARNESS_STUBS
#include <cstdio>
#include "include/libiop_amd.h"
#endif
int main(int argc, char **argv)
{
    ::testing::InitGoogleTest(&argc, argv);
#ifdef HARNESS_STUBS
    if (iopx_init(0) != IOPX_OK || iopx_profile_begin() != IOPX_OK) { std::fprintf(stderr, "kernel library: %s\n", iopx_last_error()); return 3; }
#endif
    const int rc = RUN_ALL_TESTS();
#ifdef HARNESS_STUBS
    {   // which kernels of the library the reference's tests launched through the stubs: "<kernel> <launches> ..." per line
        static char buf[1 << 16];
        if (iopx_profile_report(buf, sizeof buf) == IOPX_OK) std::printf("[ KERNELS  ]\n%s[ /KERNELS ]\n", buf);
    }
#endif
    return rc;
}
