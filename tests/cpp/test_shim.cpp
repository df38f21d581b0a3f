// Drives the C++ mirror of the reference's template API (libiop_amd/cpp/libiop_amd.hpp) the way the
// reference's own tests drive libiop (libiop/tests/algebra/test_fft.cpp:27-52,
// tests/protocols/test_fri_aux.cpp:16-45, tests/bcs/test_merkle_tree.cpp:47-94), with FieldT = the
// oracle's gf192 (same 24-byte layout as libff::gf192) and the oracle as the checker.
//   test_shim nodevice : every operator must throw std::runtime_error on a host without a GPU
//   test_shim gpu      : parity on a real MI355X
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../../libiop_amd/cpp/libiop_amd.hpp"
#include "../../oracle/field.hpp"
#include "../../oracle/algebra.hpp"
#include "../../oracle/fri.hpp"
#include "../../oracle/merkle.hpp"
#include "../../oracle/mult.hpp"
#include "../../oracle/ldt.hpp"
#include "../../oracle/pow.hpp"

typedef oracle::gf192 FieldT;

static std::mt19937_64 rng(12345);
static FieldT rnd() { FieldT r; for (int i = 0; i < 3; ++i) r.w[i] = rng(); return r; }
static std::vector<FieldT> rnd_vec(size_t n) { std::vector<FieldT> v; for (size_t i = 0; i < n; ++i) v.push_back(rnd()); return v; }

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static int run_nodevice()
{
    const libiop_amd::affine_subspace<FieldT> dom = libiop_amd::affine_subspace<FieldT>::shifted_standard_basis(3, FieldT(0));
    bool threw = false;
    try { libiop_amd::additive_FFT<FieldT>(rnd_vec(8), dom); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { libiop_amd::merkle_tree<FieldT> t(3); } catch (const std::invalid_argument &) { threw = true; }   // merkle_tree.tcc:27-31
    CHECK(threw);
    libiop_amd::merkle_tree<FieldT> t(4);
    threw = false;
    try { t.get_root(); } catch (const std::logic_error &) { threw = true; }                                 // merkle_tree.tcc:234-237
    CHECK(threw);
    printf("nodevice ok\n");
    return 0;
}

static int run_gpu()
{
    for (size_t m = 1; m <= 11; ++m) {       // test_fft.cpp:27-52 (standard basis + random shift)
        const FieldT shift = rnd();
        const libiop_amd::field_subset<FieldT> domain(libiop_amd::affine_subspace<FieldT>::shifted_standard_basis(m, shift));
        const oracle::affine_subspace<FieldT> odom = oracle::affine_subspace<FieldT>::standard(m, shift);
        const std::vector<FieldT> coeffs = rnd_vec((size_t)1 << m);
        const std::vector<FieldT> got = libiop_amd::FFT_over_field_subset<FieldT>(coeffs, domain);
        CHECK(got == oracle::naive_FFT<FieldT>(coeffs, odom.all_elements()));
        CHECK(libiop_amd::IFFT_over_field_subset<FieldT>(got, domain) == coeffs);
    }
    {   // test_fri_aux.cpp:16-45: fold of a degree<4 polynomial is P(x) on every coset
        const size_t m = 12, cs = 4;
        const FieldT shift = rnd(), x = rnd();
        const libiop_amd::field_subset<FieldT> domain(libiop_amd::affine_subspace<FieldT>::shifted_standard_basis(m, shift));
        const std::vector<FieldT> poly = rnd_vec(cs);
        auto evals = std::make_shared<std::vector<FieldT>>(libiop_amd::FFT_over_field_subset<FieldT>(poly, domain));
        FieldT px = FieldT::zero();
        for (size_t i = cs; i--; ) { px *= x; px += poly[i]; }
        auto next = libiop_amd::evaluate_next_f_i_over_entire_domain<FieldT>(evals, domain, cs, x);
        CHECK(next->size() == domain.num_elements() / cs);
        for (const FieldT &v : *next) CHECK(v == px);
    }
    {   // Merkle root == oracle's root; double construct throws (merkle_tree.tcc:98-101)
        const size_t n = 1 << 10, cs = 2;
        std::vector<std::shared_ptr<std::vector<FieldT>>> cols;
        std::vector<const uint8_t *> raw;
        for (int k = 0; k < 4; ++k) { cols.push_back(std::make_shared<std::vector<FieldT>>(rnd_vec(n))); raw.push_back((const uint8_t *)cols.back()->data()); }
        libiop_amd::merkle_tree<FieldT> tree(n / cs);
        tree.construct_with_leaves_serialized_by_cosets(cols, cs);
        std::vector<uint8_t> nodes((2 * (n / cs) - 1) * 32);
        oracle::merkle_build(raw.data(), raw.size(), sizeof(FieldT), n, cs, true, nullptr, 0, nodes.data());
        CHECK(tree.get_root() == std::string((const char *)nodes.data(), 32));
        bool threw = false;
        try { tree.construct_with_leaves_serialized_by_cosets(cols, cs); } catch (const std::logic_error &) { threw = true; }
        CHECK(threw);
    }
    {   // test_fft.cpp:88-121: multiplicative coset FFT == naive, IFFT inverts (edwards_Fr, shift = generator)
        typedef oracle::edwards_Fr Fr;
        for (size_t dim = 1; dim <= 10; ++dim) {
            const size_t n = (size_t)1 << dim;
            const oracle::mult_coset<Fr> od(n, Fr::multiplicative_generator());
            const libiop_amd::multiplicative_coset<Fr> dom(n, od.g, od.shift);
            std::vector<Fr> coeffs;
            for (size_t i = 0; i < n - (dim > 2 ? 3 : 0); ++i) { uint64_t c[3] = { rng(), rng(), rng() & 0xfffffffffull }; coeffs.push_back(Fr::from_canonical(c)); }
            const std::vector<Fr> got = libiop_amd::multiplicative_FFT<Fr>(coeffs, dom);
            CHECK(got == oracle::naive_FFT<Fr>(coeffs, od.all_elements()));
            std::vector<Fr> back = libiop_amd::multiplicative_IFFT<Fr>(got, dom);
            back.resize(coeffs.size());
            CHECK(back == coeffs);
        }
    }
    {   // tests/bcs/test_merkle_tree.cpp:127-167: every queried subset verifies against the root (zk and non-zk)
        const size_t n = 64;
        std::vector<std::shared_ptr<std::vector<FieldT>>> cols = { std::make_shared<std::vector<FieldT>>(rnd_vec(n)) };
        libiop_amd::merkle_tree<FieldT> tree(n);
        tree.construct(cols);
        const uint8_t *raw[1] = { (const uint8_t *)cols[0]->data() };
        std::vector<uint8_t> nodes((2 * n - 1) * 32);
        oracle::merkle_build(raw, 1, sizeof(FieldT), n, 1, true, nullptr, 0, nodes.data());
        for (uint64_t subset : { 0x1ull, 0x8000000000000000ull, 0xF0F0ull, 0x123456789ABCDEFull, ~0ull }) {
            std::vector<size_t> pos;
            for (size_t k = 0; k < n; ++k) if (subset >> k & 1) pos.push_back(k);
            const libiop_amd::merkle_tree_set_membership_proof mp = tree.get_set_membership_proof(pos);
            const std::vector<size_t> idx = oracle::membership_proof_node_indices(n, pos);
            CHECK(mp.auxiliary_hashes.size() == idx.size());
            std::vector<std::vector<uint8_t>> leaf_hashes, aux;
            for (size_t i = 0; i < idx.size(); ++i) {
                CHECK(mp.auxiliary_hashes[i] == std::string((const char *)&nodes[32 * idx[i]], 32));
                aux.emplace_back(mp.auxiliary_hashes[i].begin(), mp.auxiliary_hashes[i].end());
            }
            for (size_t q : pos) leaf_hashes.emplace_back(nodes.begin() + 32 * (n - 1 + q), nodes.begin() + 32 * (n + q));
            CHECK(oracle::membership_proof_validate(nodes.data(), n, pos, leaf_hashes, aux));
        }
    }
    {   // tests/protocols/test_ldt_reducer.cpp: the combined oracle equals the oracle's literal combination
        const size_t m = 9;
        const FieldT shift = rnd();
        const libiop_amd::field_subset<FieldT> domain(libiop_amd::affine_subspace<FieldT>::shifted_standard_basis(m, shift));
        const std::vector<size_t> degrees = { 200, 77, 200, 199, 1 };
        std::vector<std::shared_ptr<std::vector<FieldT>>> evals;
        std::vector<std::vector<FieldT>> oevals;
        for (size_t k = 0; k < degrees.size(); ++k) { oevals.push_back(rnd_vec((size_t)1 << m)); evals.push_back(std::make_shared<std::vector<FieldT>>(oevals.back())); }
        const std::vector<FieldT> r = rnd_vec(2 * degrees.size());
        libiop_amd::combined_LDT_virtual_oracle<FieldT> vo(domain, degrees);
        vo.set_random_coefficients(r);
        oracle::combined_LDT_virtual_oracle<FieldT> ovo(degrees);
        ovo.set_random_coefficients(r);
        CHECK(*vo.evaluated_contents(evals) == ovo.evaluated_contents(oracle::affine_subspace<FieldT>::standard(m, shift), oevals));
        bool threw = false;
        try { vo.set_random_coefficients(rnd_vec(3)); } catch (const std::invalid_argument &) { threw = true; }     // ldt_reducer_aux.tcc:29-32
        CHECK(threw);
    }
    {   // tests/snark/test_pow.cpp:13-33
        const libiop_amd::pow_parameters params(20, 1);
        const libiop_amd::binary_pow prover(params, 32);
        const std::string challenge = "abcdefghijklmnopqrstuvwxyzabcdef";
        const std::string proof = prover.solve_pow(challenge);
        CHECK(oracle::pow_verify_blake2b((const uint8_t *)challenge.data(), (const uint8_t *)proof.data(), params.pow_bitlen()));
        uint8_t want[32];
        oracle::pow_solve_blake2b((const uint8_t *)challenge.data(), params.pow_bitlen(), want);
        CHECK(proof == std::string((const char *)want, 32));
    }
    printf("gpu ok\n");
    return 0;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "nodevice";
    return mode == "gpu" ? run_gpu() : run_nodevice();
}
