"""The reference-side stubs of INTEGRATION.md, compiled and run.

This file needs nothing of the reference tree (it runs wherever the CPU suite runs; tests/test_reference_harness.py compiles the same text into libiop itself
where /root/reference exists).  The stubs — explicit specialisations
of libiop's function templates for libff::gf192 / libff::edwards_Fr — are compiled VERBATIM (code blocks cut out of INTEGRATION.md, minus the two
#include lines) against the nearest thing that can be: `namespace libiop` holding the mirror classes of libiop_amd/cpp/libiop_amd.hpp, whose member
declarations tests/test_reference_signatures.py compares with the reference's header text, and the reference's primary templates DECLARED with the
reference's signatures (fft.hpp:28-52, fri_aux.tcc:36-41,106-111); libff::gf192 / edwards_Fr = the plain 24-byte types of cpp/fields.hpp.  The program
then calls every stub and compares with the mirror's own functions on the CPU build of the kernels.  What this shows: the documented binding is
well-formed against classes with the reference's declarations (an `&domain.shift()` does not compile here: shift() is a prvalue) and forwards correctly.
libiop's own code compiled against the stubs: tests/harness."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PREAMBLE = r'''
#include "%(root)s/libiop_amd/cpp/libiop_amd.hpp"
#include "%(root)s/libiop_amd/cpp/fields.hpp"
#include <cstdio>
#include <cstdlib>
namespace libff { typedef libiop_amd::gf192_element gf192; typedef libiop_amd::edwards_Fr_element edwards_Fr; }
namespace libiop {
using libiop_amd::affine_subspace; using libiop_amd::multiplicative_coset; using libiop_amd::field_subset;
using libiop_amd::affine_subspace_type; using libiop_amd::multiplicative_coset_type;
// the reference's primary templates, declared with its signatures (never defined here: only the stubs' specialisations exist)
template<typename FieldT> std::vector<FieldT> additive_FFT(const std::vector<FieldT> &poly_coeffs, const affine_subspace<FieldT> &domain);            // fft.hpp:28-30
template<typename FieldT> std::vector<FieldT> additive_IFFT(const std::vector<FieldT> &evals, const affine_subspace<FieldT> &domain);                  // fft.hpp:32-34
template<typename FieldT> std::vector<FieldT> multiplicative_FFT(const std::vector<FieldT> &poly_coeffs, const multiplicative_coset<FieldT> &domain);  // fft.hpp:46-48
template<typename FieldT> std::vector<FieldT> multiplicative_IFFT(const std::vector<FieldT> &evals, const multiplicative_coset<FieldT> &domain);       // fft.hpp:50-52
template<typename FieldT> std::shared_ptr<std::vector<FieldT>> additive_evaluate_next_f_i_over_entire_domain(                                          // fri_aux.tcc:36-41
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const field_subset<FieldT> &f_i_domain, const size_t coset_size, const FieldT x_i);
template<typename FieldT> std::shared_ptr<std::vector<FieldT>> multiplicative_evaluate_next_f_i_over_entire_domain(                                    // fri_aux.tcc:106-111
    const std::shared_ptr<std::vector<FieldT>> &f_i_evals, const field_subset<FieldT> &f_i_domain, const size_t coset_size, const FieldT x_i);
}
'''

MAIN = r'''
template<typename F> static std::vector<F> seeded(uint64_t seed, size_t n, bool prime)
{
    std::vector<F> v(n);
    uint64_t x = seed;
    for (size_t i = 0; i < n; ++i) {
        if (prime) { x = x * 6364136223846793005ull + 1442695040888963407ull; v[i] = F(x >> 8); }
        else for (int k = 0; k < 3; ++k) { x = x * 6364136223846793005ull + 1442695040888963407ull; v[i].w[k] = x; }
    }
    return v;
}
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED line %%d: %%s\n", __LINE__, #c); return 1; } } while (0)
int main()
{
    using namespace libiop_amd;
    if (iopx_init(0) != IOPX_OK) { std::printf("no device: %%s\n", iopx_last_error()); return 2; }
    {   // GF(2^192): the stubs against the mirror's own functions
        typedef libff::gf192 F;
        const F shift((uint64_t)1 << 7);
        const affine_subspace<F> S = affine_subspace<F>::shifted_standard_basis(7, shift);
        const std::vector<F> coeffs = seeded<F>(11, 100, false);
        const std::vector<F> evals = libiop::additive_FFT<F>(coeffs, S);
        REQUIRE(evals == libiop_amd::additive_FFT<F>(coeffs, S));
        std::vector<F> padded = coeffs; padded.resize(128, F(0));
        REQUIRE(libiop::additive_IFFT<F>(evals, S) == padded);
        const auto f = std::make_shared<std::vector<F>>(evals);
        const F x = seeded<F>(12, 1, false)[0];
        const field_subset<F> D(S);
        REQUIRE(*libiop::additive_evaluate_next_f_i_over_entire_domain<F>(f, D, 4, x) == *libiop_amd::evaluate_next_f_i_over_entire_domain<F>(f, D, 4, x));
    }
    {   // the 181-bit prime field
        typedef libff::edwards_Fr F;
        const field_subset<F> unshifted((size_t)1 << 7);
        const field_subset<F> D((size_t)1 << 7, unshifted.element_outside_of_subset());
        const multiplicative_coset<F> C = D.coset();
        const std::vector<F> coeffs = seeded<F>(13, 77, true);
        const std::vector<F> evals = libiop::multiplicative_FFT<F>(coeffs, C);
        REQUIRE(evals == libiop_amd::multiplicative_FFT<F>(coeffs, C));
        std::vector<F> padded = coeffs; padded.resize(128, F(0));
        REQUIRE(libiop::multiplicative_IFFT<F>(evals, C) == padded);
        const auto f = std::make_shared<std::vector<F>>(evals);
        const F x = seeded<F>(14, 1, true)[0];
        REQUIRE(*libiop::multiplicative_evaluate_next_f_i_over_entire_domain<F>(f, D, 2, x) == *libiop_amd::evaluate_next_f_i_over_entire_domain<F>(f, D, 2, x));
    }
    std::printf("stubs ok\n");
    return 0;
}
'''


def _blocks():
    """The cpp code blocks of INTEGRATION.md that follow the three headings whose stubs are complete function specialisations."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    out = {}
    for key, heading in (("fft", "### FFT / IFFT"), ("fold", "### FRI fold"), ("prime", "### Prime-field (multiplicative-coset) transforms")):
        at = text.index(heading)
        m = re.compile(r"```cpp\n(.*?)```", re.S).search(text, at)
        out[key] = m.group(1)
    return out


def test_the_documented_stubs_compile_and_forward_correctly(tmp_path):
    from emu_lib import emu
    emu()
    b = _blocks()
    fft = "\n".join(line for line in b["fft"].splitlines() if not line.startswith("#include"))          # the two #include lines name libiop's build tree
    assert "template<>" in fft and "namespace libiop" in fft
    src = PREAMBLE % {"root": ROOT} + fft + "\nnamespace libiop {\n" + b["fold"] + "\n" + b["prime"] + "\n}\n" + MAIN.replace("%%", "%")
    cpp = tmp_path / "stubs.cpp"
    cpp.write_text(src)
    emu_dir = os.path.join(ROOT, "tests", "emu")
    exe = str(tmp_path / "stubs")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror=return-type", str(cpp), "-o", exe, os.path.join(emu_dir, "libiopx_emu.so"), "-Wl,-rpath," + emu_dir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "stubs ok" in r.stdout, r.stdout + r.stderr


def test_a_stub_that_takes_the_address_of_shift_does_not_compile(tmp_path):
    """Round 5's documented stub (`(const uint64_t*)&domain.shift()`) against the same declarations: ill-formed, as it is against the reference."""
    bad = PREAMBLE % {"root": ROOT} + r'''
namespace libiop {
template<> std::vector<libff::gf192> additive_FFT<libff::gf192>(const std::vector<libff::gf192> &poly_coeffs, const affine_subspace<libff::gf192> &domain)
{
    std::vector<libff::gf192> out(domain.num_elements());
    libiop_amd::check(iopx_add_fft_gf192((const uint64_t*)poly_coeffs.data(), poly_coeffs.size(), (const uint64_t*)domain.basis().data(), domain.dimension(),
                                         (const uint64_t*)&domain.shift(), (uint64_t*)out.data()));
    return out;
}
}
int main() { return 0; }
'''
    cpp = tmp_path / "bad.cpp"
    cpp.write_text(bad)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", str(cpp)], capture_output=True, text=True)
    assert r.returncode != 0 and ("rvalue" in r.stderr or "temporary" in r.stderr), r.stderr[-2000:]
