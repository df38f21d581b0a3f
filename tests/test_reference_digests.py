"""The oracle and the native provers (CPU build of the kernel sources) against transcripts of libiop's OWN prover: tests/golden/reference_over_shim.json,
generated in the build container by tests/harness (the reference's sources compiled unmodified over a stand-in libff).  Runs everywhere — the fixture is data."""
import pytest

import emu_lib
import reference_digest_cases as rc


@pytest.mark.parametrize("e", rc.entries(), ids=rc.ident)
def test_oracle_equals_the_references_own_prover(e):
    rc.check_oracle(e)


@pytest.mark.parametrize("e", [e for e in rc.entries() if e["log_n"] <= 8], ids=rc.ident)
def test_native_prover_on_cpu_kernels_equals_the_references_own_prover(e):
    rc.check_native(emu_lib.emu(), e)
