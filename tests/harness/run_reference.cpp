// TEST INFRASTRUCTURE ONLY — container-only harness (SURVEY §8(b) last row).
// Compiles the reference's OWN Aurora prover and verifier straight from /root/reference (nothing copied) over the stand-in libff of tests/harness/shim,
// runs prover -> verifier (Aurora; Fractal with its indexer; Ligero) on the seeded instances the parity tests use, and writes the transcript in
// the byte form of oracle::bcs_transcript::serialize.  Not the FRI-only SNARK: the reference's dummy_oracle::evaluated_contents returns an EMPTY vector
// (protocols/encoded/dummy_protocol.tcc:24-29 reserves, then loops to size()), so FRI_snark_prover folds out of bounds — it crashes here.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <libff/algebra/fields/binary/gf192.hpp>
#include <libff/algebra/curves/edwards/edwards_pp.hpp>
#include "libiop/snark/aurora_snark.hpp"
#include "libiop/snark/fractal_snark.hpp"
#include "libiop/snark/ligero_snark.hpp"
#include "libiop/relations/examples/r1cs_examples.hpp"
#include "libiop/bcs/common_bcs_parameters.hpp"

#ifdef HARNESS_STUBS
#include "libiop/bcs/pow.hpp"
#include "libiop/protocols/ldt/ldt_reducer_aux.hpp"
#include "libiop/protocols/encoded/common/rowcheck.hpp"
#include "libiop/protocols/encoded/r1cs_rs_iop/r1cs_rs_iop.hpp"
#include "libiop/protocols/encoded/sumcheck/sumcheck.hpp"
#include "libiop/protocols/encoded/lincheck/basic_lincheck_aux.hpp"
#include "stubs.inc"                    // the stub definitions of INTEGRATION.md, verbatim (tests/harness/make_shadow.py)
#endif

using namespace libiop;


template<typename FieldT>
static std::vector<uint8_t> canonical_bytes(const bcs_transformation_transcript<FieldT, binary_hash_digest> &t)
{
    std::vector<uint8_t> out;
    auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(v >> (8 * i))); };
    auto raw = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; out.insert(out.end(), b, b + n); };
    u64(t.prover_messages_.size());
    for (auto &m : t.prover_messages_) { u64(m.size()); raw(m.data(), m.size() * sizeof(FieldT)); }
    u64(t.MT_roots_.size());
    for (auto &r : t.MT_roots_) raw(r.data(), r.size());
    for (size_t k = 0; k < t.query_positions_.size(); ++k) {
        u64(t.query_positions_[k].size());
        for (size_t p : t.query_positions_[k]) u64(p);
        u64(t.MT_leaf_positions_[k].size());
        for (size_t p : t.MT_leaf_positions_[k]) u64(p);
        u64(t.query_responses_[k].empty() ? 0 : t.query_responses_[k][0].size());
        for (auto &col : t.query_responses_[k]) raw(col.data(), col.size() * sizeof(FieldT));
        u64(t.MT_set_membership_proofs_[k].auxiliary_hashes.size());
        for (auto &h : t.MT_set_membership_proofs_[k].auxiliary_hashes) raw(h.data(), h.size());
    }
    raw(t.proof_of_work_.data(), t.proof_of_work_.size());
    return out;
}

static std::string hex(const std::string &bytes)
{
    static const char *d = "0123456789abcdef";
    std::string out;
    for (unsigned char c : bytes) { out.push_back(d[c >> 4]); out.push_back(d[c & 15]); }
    return out;
}

struct job {
    std::string protocol;                 // aurora | fractal | ligero
    size_t log_n, k, rs_extra, localization;
    uint64_t seed;
    const char *out_path;
};

static void begin_kernel_count()
{
#ifdef HARNESS_STUBS
    if (iopx_profile_begin() != IOPX_OK) { fprintf(stderr, "kernel library: %s\n", iopx_last_error()); exit(3); }
#endif
}
// which kernels of the library ran since begin_kernel_count, and how often: "<kernel> <launches> <ms> <bytes> <products>" per line
static std::string end_kernel_count()
{
    std::string json = "{";
#ifdef HARNESS_STUBS
    std::vector<char> buf(1 << 16);
    if (iopx_profile_report(buf.data(), buf.size()) != IOPX_OK) { fprintf(stderr, "kernel library: %s\n", iopx_last_error()); exit(3); }
    std::istringstream in(buf.data());
    std::string name, rest;
    unsigned long count;
    while (in >> name >> count && std::getline(in, rest)) json += (json.size() > 1 ? ", \"" : "\"") + name + "\": " + std::to_string(count);
#endif
    return json + "}";
}

template<typename FieldT>
static int run(const field_subset_type domain_type, const job &j)
{
    const LDT_reducer_soundness_type ldt = LDT_reducer_soundness_type::optimistic_heuristic;       // the instrument programs' defaults
    const FRI_soundness_type fri = FRI_soundness_type::heuristic;
    const size_t n = (size_t)1 << j.log_n, m = n - 1;
    std::vector<uint8_t> bytes;
    std::string launches, roots = "[";
    bool ok = false;
    FieldT::seed_random(j.seed);
    if (j.protocol == "aurora") {
        r1cs_example<FieldT> example = generate_r1cs_example<FieldT>(n, j.k, m);          // the reference's own generator over the seeded stream
        aurora_snark_parameters<FieldT, binary_hash_digest> parameters(128, ldt, fri, blake2b_type, j.localization, j.rs_extra, false, domain_type,
                                                                       example.constraint_system_.num_constraints(), example.constraint_system_.num_variables());
        begin_kernel_count();
        const aurora_snark_argument<FieldT, binary_hash_digest> proof =
            aurora_snark_prover<FieldT, binary_hash_digest>(example.constraint_system_, example.primary_input_, example.auxiliary_input_, parameters);
        launches = end_kernel_count();
        ok = aurora_snark_verifier<FieldT, binary_hash_digest>(example.constraint_system_, example.primary_input_, proof, parameters);
        bytes = canonical_bytes<FieldT>(proof);
    } else if (j.protocol == "fractal") {
        r1cs_example<FieldT> example = generate_r1cs_example<FieldT>(n, j.k, m);
        auto make_parameters = [&] {
            return fractal_snark_parameters<FieldT, binary_hash_digest>(128, ldt, fri, blake2b_type, j.localization, j.rs_extra, false, domain_type,
                                                                        std::make_shared<r1cs_constraint_system<FieldT>>(example.constraint_system_));
        };
        fractal_snark_parameters<FieldT, binary_hash_digest> parameters = make_parameters();
        begin_kernel_count();
        std::pair<bcs_prover_index<FieldT, binary_hash_digest>, bcs_verifier_index<FieldT, binary_hash_digest>> index = fractal_snark_indexer(parameters);
        const fractal_snark_argument<FieldT, binary_hash_digest> proof = fractal_snark_prover(index.first, example.primary_input_, example.auxiliary_input_, parameters);
        launches = end_kernel_count();
        parameters = make_parameters();                                                  // as instrument_fractal_snark.cpp:178-187 does before verifying
        ok = fractal_snark_verifier<FieldT, binary_hash_digest>(index.second, example.primary_input_, proof, parameters);
        bytes = canonical_bytes<FieldT>(proof);
        for (auto &r : index.second.index_MT_roots_) roots += (roots.size() > 1 ? ", \"" : "\"") + hex(r) + "\"";
    } else if (j.protocol == "ligero") {                                                 // instrument_ligero_snark.cpp:65-130, non-zk; no oracle restates Ligero:
        r1cs_example<FieldT> example = generate_r1cs_example<FieldT>(n, j.k, m);         // the plain and the stubbed program must agree with each other
        ligero_snark_parameters<FieldT, binary_hash_digest> parameters;
        parameters.security_level_ = 128;
        parameters.LDT_reducer_soundness_type_ = ldt;
        parameters.height_width_ratio_ = 0.1;
        parameters.RS_extra_dimensions_ = j.rs_extra;
        parameters.make_zk_ = false;
        parameters.domain_type_ = domain_type;
        parameters.bcs_params_ = default_bcs_params<FieldT, binary_hash_digest>(blake2b_type, 128, j.log_n);
        begin_kernel_count();
        const ligero_snark_argument<FieldT, binary_hash_digest> proof =
            ligero_snark_prover<FieldT, binary_hash_digest>(example.constraint_system_, example.primary_input_, example.auxiliary_input_, parameters);
        launches = end_kernel_count();
        ok = ligero_snark_verifier<FieldT, binary_hash_digest>(example.constraint_system_, example.primary_input_, proof, parameters);
        bytes = canonical_bytes<FieldT>(proof);
    } else {
        fprintf(stderr, "unknown protocol %s\n", j.protocol.c_str());
        return 2;
    }
    if (j.out_path) { std::ofstream f(j.out_path, std::ios::binary); f.write((const char *)bytes.data(), (std::streamsize)bytes.size()); }
    printf("\n{\"verifier_accepts\": %s, \"transcript_bytes\": %zu, \"index_roots\": %s], \"kernel_launches_in_prover\": %s}\n", ok ? "true" : "false", bytes.size(),
           roots.c_str(), launches.c_str());
    return ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: %s aurora|fractal|ligero gf192|edwards_Fr log_n num_inputs seed rs_extra localization out.bin\n", argv[0]); return 2; }
    job j;
    j.protocol = argv[1];
    const std::string field = argv[2];
    j.log_n = strtoul(argv[3], 0, 0);
    j.k = strtoul(argv[4], 0, 0); j.seed = strtoull(argv[5], 0, 0); j.rs_extra = strtoul(argv[6], 0, 0); j.localization = strtoul(argv[7], 0, 0);
    j.out_path = argv[8];
#ifdef HARNESS_STUBS
    if (iopx_init(0) != IOPX_OK) { fprintf(stderr, "kernel library: %s\n", iopx_last_error()); return 3; }
#endif
    if (field == "gf192") return run<libff::gf192>(affine_subspace_type, j);
    if (field == "edwards_Fr") {
        libff::edwards_pp::init_public_params();
        return run<libff::edwards_Fr>(multiplicative_coset_type, j);
    }
    fprintf(stderr, "unknown field %s\n", field.c_str());
    return 2;
}
