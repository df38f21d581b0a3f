// Aurora prover bench through the C++ surface (libiop_amd/cpp/aurora.hpp): the BASELINE workload — generate_r1cs_example(n, 15, n - 1)
// over GF(2^192) (or --field edwards), security 128, RS_extra_dimensions 5, localization 2, non-zk, BLAKE2b — one complete proof per
// step with instance and witness resident in HBM.  Prints one JSON line: ms per proof, a BLAKE2b-256 of the transcript bytes (the GPU
// test compares it with the Python prover's and the oracle's), PCIe bytes per proof.
//   g++ -O2 -std=c++17 tools/cpp/aurora_bench.cpp -o tools/cpp/aurora_bench -Llibiop_amd/lib -liop_amd -Wl,-rpath,$PWD/libiop_amd/lib
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../libiop_amd/cpp/aurora.hpp"
#include "../../libiop_amd/cpp/fields.hpp"

using namespace libiop_amd;

template<typename F>
static int run(size_t log_n, int steps, int warmup, uint64_t seed)
{
    const size_t n = (size_t)1 << log_n;
    r1cs_example<F> ex = generate_r1cs_example<F>(n, 15, n - 1, seed);
    const aurora_snark_parameters<F> params(n, n - 1, 15);
    ex.constraint_system.prepare_device();
    std::vector<F> z(1, field_host<F>::one());
    z.insert(z.end(), ex.primary_input.begin(), ex.primary_input.end());
    z.insert(z.end(), ex.auxiliary_input.begin(), ex.auxiliary_input.end());
    const device_vector<F> d_z(device_array<F>::from_host(z));
    std::string transcript;
    for (int i = 0; i < warmup; ++i) transcript = aurora_snark_prover<F>(ex.constraint_system, ex.primary_input, ex.auxiliary_input, params, &d_z).serialize();
    check(iopx_synchronize());
    check(iopx_transfer_stats(nullptr, nullptr, 1));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) transcript = aurora_snark_prover<F>(ex.constraint_system, ex.primary_input, ex.auxiliary_input, params, &d_z).serialize();
    check(iopx_synchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    uint64_t h2d = 0, d2h = 0;
    check(iopx_transfer_stats(&h2d, &d2h, 0));
    uint8_t digest[32];
    check(iopx_blake2b_host(digest, 32, transcript.data(), transcript.size(), nullptr, 0));
    char hex[65];
    for (int i = 0; i < 32; ++i) snprintf(hex + 2 * i, 3, "%02x", digest[i]);
    printf("{\"prover\": \"libiop_amd/cpp/aurora.hpp\", \"field\": \"%s\", \"log_n\": %zu, \"steps\": %d, \"ms_per_proof\": %.3f, \"argument_bytes\": %zu, "
           "\"transcript_blake2b\": \"%s\", \"pcie_h2d_bytes_per_proof\": %llu, \"pcie_d2h_bytes_per_proof\": %llu}\n",
           field_host<F>::additive() ? "gf192" : "edwards_Fr", log_n, steps, ms, transcript.size(), hex, (unsigned long long)(h2d / steps), (unsigned long long)(d2h / steps));
    return 0;
}

int main(int argc, char **argv)
{
    size_t log_n = 20;
    int steps = 5, warmup = 2;
    std::string field = "gf192";
    uint64_t seed = 0x2204;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--log-n") log_n = (size_t)atoi(argv[i + 1]);
        else if (k == "--steps") steps = atoi(argv[i + 1]);
        else if (k == "--warmup") warmup = atoi(argv[i + 1]);
        else if (k == "--field") field = argv[i + 1];
        else if (k == "--seed") seed = strtoull(argv[i + 1], nullptr, 0);
    }
    try {
        check(iopx_init(0));
        return field == "edwards" ? run<edwards_Fr_element>(log_n, steps, warmup, seed) : run<gf192_element>(log_n, steps, warmup, seed);
    } catch (const std::exception &e) {
        fprintf(stderr, "aurora_bench: %s\n", e.what());
        return 1;
    }
}
