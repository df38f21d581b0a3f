"""End-to-end FRI (prover through the C ABI on the CPU build of the kernels, independent oracle verifier)."""
import pytest

import fri_cases as fc
from emu_lib import emu


@pytest.mark.parametrize("m,rs_extra,loc_param,queries,pow_bits,kind", [(8, 2, 2, 6, 6, "standard"), (10, 3, 2, 10, 9, "random"), (7, 2, 1, 4, 0, "random"),
                                                                       (9, 2, 3, 8, 10, "standard")])
def test_prove_and_verify(m, rs_extra, loc_param, queries, pow_bits, kind):
    torch, to_device = fc.host_env()
    assert fc.prove_and_verify(emu(), torch, to_device, m, rs_extra, loc_param, queries, pow_bits, 5, kind)


@pytest.mark.parametrize("log_n,rs_extra,loc_param,queries,pow_bits", [(8, 2, 2, 6, 5), (9, 3, 1, 8, 7), (10, 2, 3, 10, 0)])
def test_prove_and_verify_multiplicative(log_n, rs_extra, loc_param, queries, pow_bits):
    torch, to_device = fc.host_env()
    assert fc.prove_and_verify_multiplicative(emu(), torch, to_device, log_n, rs_extra, loc_param, queries, pow_bits, 7)
