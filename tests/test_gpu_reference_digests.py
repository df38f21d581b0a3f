"""The native HIP provers, through the C ABI on the MI355X, against transcripts of libiop's OWN prover (tests/golden/reference_over_shim.json: the
reference's sources compiled unmodified over a stand-in libff in the build container, tests/harness).  The fixture is data; nothing here reads /root/reference."""
import pytest

import reference_digest_cases as rc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import libiop_amd
    lib = libiop_amd.lib()          # raises if the HIP library is missing: no fallback
    lib.init(0)
    return lib


@pytest.mark.parametrize("e", rc.entries(), ids=rc.ident)
def test_hip_prover_equals_the_references_own_prover(gpu, e):
    rc.check_native(gpu, e)
