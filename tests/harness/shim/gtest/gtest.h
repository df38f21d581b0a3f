// TEST INFRASTRUCTURE ONLY (tests/harness): a stand-in for googletest (an empty submodule of the reference tree) — what libiop's test files use of it:
// TEST, EXPECT_* / ASSERT_* (streamable), EXPECT_THROW, FRIEND_TEST's class naming, a main() that runs every registered test.
#pragma once
#include <cstdio>
#include <exception>
#include <sstream>
#include <string>
#include <vector>
#include "gtest/gtest_prod.h"

namespace testing {

struct Message {
    std::ostringstream s;
    template<typename T> Message &operator<<(const T &v) { s << v; return *this; }
};
struct TestInfo { const char *suite, *name; void (*run)(); };
inline std::vector<TestInfo> &registry() { static std::vector<TestInfo> r; return r; }
inline int &failures_in_current_test() { static int n = 0; return n; }
struct Registrar { Registrar(const char *suite, const char *name, void (*run)()) { registry().push_back({ suite, name, run }); } };
struct AssertHelper {
    const char *file; int line; const char *what;
    void operator=(const Message &m) const
    {
        ++failures_in_current_test();
        std::fprintf(stderr, "%s:%d: Failure\n  %s\n  %s\n", file, line, what, m.s.str().c_str());
    }
};
class Test { public: virtual ~Test() {} virtual void TestBody() = 0; };
// --gtest_filter=-A*:B* — only the negative form (tests to leave out), ':' separated, '*' wildcards: what tests/harness/run_reftests.py needs
inline std::vector<std::string> &excluded() { static std::vector<std::string> v; return v; }
inline bool glob(const char *pat, const char *str)
{
    if (!*pat) return !*str;
    if (*pat == '*') return glob(pat + 1, str) || (*str && glob(pat, str + 1));
    return *pat == *str && glob(pat + 1, str + 1);
}
inline void InitGoogleTest(int *argc, char **argv)
{
    for (int i = 1; i < *argc; ++i) {
        const std::string a = argv[i];
        if (a.rfind("--gtest_filter=-", 0) == 0) {
            std::string rest = a.substr(16);
            size_t at;
            while ((at = rest.find(':')) != std::string::npos) { excluded().push_back(rest.substr(0, at)); rest = rest.substr(at + 1); }
            if (!rest.empty()) excluded().push_back(rest);
        }
    }
}

inline int RunAllTests()
{
    int failed = 0;
    for (auto &t : registry()) {
        const std::string full = std::string(t.suite) + "." + t.name;
        bool skip = false;
        for (auto &pat : excluded()) skip = skip || glob(pat.c_str(), full.c_str());
        if (skip) { std::printf("[ SKIPPED  ] %s\n", full.c_str()); continue; }
        failures_in_current_test() = 0;
        std::printf("[ RUN      ] %s.%s\n", t.suite, t.name);
        std::fflush(stdout);
        try { t.run(); }
        catch (const std::exception &e) { ++failures_in_current_test(); std::fprintf(stderr, "  uncaught exception: %s\n", e.what()); }
        catch (...) { ++failures_in_current_test(); std::fprintf(stderr, "  uncaught exception\n"); }
        std::printf("%s %s.%s\n", failures_in_current_test() ? "[  FAILED  ]" : "[       OK ]", t.suite, t.name);
        if (failures_in_current_test()) ++failed;
    }
    std::printf("[==========] %zu tests ran.\n[  PASSED  ] %zu tests.\n", registry().size(), registry().size() - failed);
    if (failed) std::printf("[  FAILED  ] %d tests.\n", failed);
    return failed ? 1 : 0;
}

} // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()
#define TEST(suite, name)                                                                                    \
    class suite##_##name##_Test : public ::testing::Test { public: void TestBody() override; static void Run() { suite##_##name##_Test t; t.TestBody(); } }; \
    static ::testing::Registrar suite##_##name##_registrar(#suite, #name, &suite##_##name##_Test::Run);     \
    void suite##_##name##_Test::TestBody()

#define IOPX_GTEST_BLOCKER_ switch (0) case 0: default:
#define IOPX_GTEST_FAIL_(what) ::testing::AssertHelper{ __FILE__, __LINE__, what } = ::testing::Message()
#define IOPX_GTEST_CHECK_(cond, what, on_fail) IOPX_GTEST_BLOCKER_ if (cond) ; else on_fail IOPX_GTEST_FAIL_(what)
#define EXPECT_TRUE(c)  IOPX_GTEST_CHECK_((c), "expected true: " #c, )
#define EXPECT_FALSE(c) IOPX_GTEST_CHECK_(!(c), "expected false: " #c, )
#define EXPECT_EQ(a, b) IOPX_GTEST_CHECK_((a) == (b), "expected equality of " #a " and " #b, )
#define EXPECT_NE(a, b) IOPX_GTEST_CHECK_((a) != (b), "expected " #a " != " #b, )
#define EXPECT_LE(a, b) IOPX_GTEST_CHECK_((a) <= (b), "expected " #a " <= " #b, )
#define EXPECT_LT(a, b) IOPX_GTEST_CHECK_((a) < (b), "expected " #a " < " #b, )
#define EXPECT_GE(a, b) IOPX_GTEST_CHECK_((a) >= (b), "expected " #a " >= " #b, )
#define EXPECT_GT(a, b) IOPX_GTEST_CHECK_((a) > (b), "expected " #a " > " #b, )
#define ASSERT_TRUE(c)  IOPX_GTEST_CHECK_((c), "expected true: " #c, return)
#define ASSERT_FALSE(c) IOPX_GTEST_CHECK_(!(c), "expected false: " #c, return)
#define ASSERT_EQ(a, b) IOPX_GTEST_CHECK_((a) == (b), "expected equality of " #a " and " #b, return)
#define ASSERT_NE(a, b) IOPX_GTEST_CHECK_((a) != (b), "expected " #a " != " #b, return)
#define ASSERT_LE(a, b) IOPX_GTEST_CHECK_((a) <= (b), "expected " #a " <= " #b, return)
#define ASSERT_LT(a, b) IOPX_GTEST_CHECK_((a) < (b), "expected " #a " < " #b, return)
#define ASSERT_GE(a, b) IOPX_GTEST_CHECK_((a) >= (b), "expected " #a " >= " #b, return)
#define ASSERT_GT(a, b) IOPX_GTEST_CHECK_((a) > (b), "expected " #a " > " #b, return)
#define EXPECT_THROW(statement, exception_type)                                                             \
    IOPX_GTEST_BLOCKER_ if ([&] { try { statement; } catch (const exception_type &) { return true; } catch (...) { return false; } return false; }()) ; \
    else IOPX_GTEST_FAIL_("expected " #statement " to throw " #exception_type)
