"""The oracle and the CPU build of the kernels against outputs of the REFERENCE'S OWN functions (tests/golden/reference_functions.json, generated in the build
container by tests/harness/reference_vectors.cpp: libiop's sources compiled unmodified over a stand-in libff) — 578 seeded cases of SURVEY §8 rows A1 – A7,
B3 – B4, C1 – C2, f2, f3: additive FFT / IFFT / known-degree IFFT over standard, shifted and seeded bases, with fewer coefficients than points; the
multiplicative ones incl. non-power-of-two lengths and seeded shifts; folds with coset sizes 2, 4, 8 and a challenge inside the domain; the LDT combination with
submaximal degrees; trees of 1 – 4 oracles with cosets of 1 – 8 over both position maps; the grind.  Runs everywhere: the fixture is data."""
import pytest

import emu_lib
import reference_function_cases as fc

GROUPS = sorted({e["case"] for e in fc.entries()})


@pytest.mark.parametrize("case", GROUPS)
def test_oracle_equals_the_references_own_functions(case):
    es = fc.entries({case})
    assert es
    for e in es:
        fc.check(e)


@pytest.mark.parametrize("case", GROUPS)
def test_cpu_built_kernels_equal_the_references_own_functions(case):
    lib = emu_lib.emu()
    for e in fc.entries({case}):
        fc.check(e, lib)
