"""The HIP kernels, through the C ABI on the MI355X, against outputs of the REFERENCE'S OWN functions (tests/golden/reference_functions.json; see
tests/test_reference_functions.py).  The fixture is data; nothing here reads /root/reference."""
import pytest

import reference_function_cases as fc

pytestmark = pytest.mark.gpu
GROUPS = sorted({e["case"] for e in fc.entries()})


@pytest.fixture(scope="module")
def gpu():
    import libiop_amd
    lib = libiop_amd.lib()          # raises if the HIP library is missing: no fallback
    lib.init(0)
    return lib


@pytest.mark.parametrize("case", GROUPS)
def test_hip_kernels_equal_the_references_own_functions(gpu, case):
    for e in fc.entries({case}):
        fc.check(e, gpu)
