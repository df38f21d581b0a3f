"""FRI-only SNARK end to end (prover through the C ABI on the CPU build of the kernels): transcripts byte-equal to the oracle's
independent prover, accepted by its verifier; tampering and high-degree codewords rejected."""
import pytest
import torch

import fri_cases as fc
from emu_lib import emu

CPU = torch.device("cpu")


@pytest.mark.parametrize("field_name,dim,rs_extra,loc_param,interactions,queries", [
    ("gf192", 8, 2, 2, 1, 6), ("gf192", 10, 3, 2, 1, 10), ("gf192", 7, 2, 1, 2, 4), ("gf192", 9, 2, 3, 1, 8),
    ("edwards_Fr", 8, 2, 2, 1, 6), ("edwards_Fr", 9, 3, 1, 1, 8), ("edwards_Fr", 10, 2, 3, 2, 10)])
def test_fri_snark(field_name, dim, rs_extra, loc_param, interactions, queries):
    assert fc.prove_and_verify(emu(), torch, CPU, field_name, dim, rs_extra, loc_param, interactions, queries, 5)


@pytest.mark.parametrize("field_name,dim,rs_extra,loc_param,interactions,queries", [
    ("gf192", 8, 2, 2, 1, 6), ("gf192", 10, 3, 2, 1, 10), ("gf192", 7, 2, 1, 2, 4), ("edwards_Fr", 8, 2, 2, 1, 6), ("edwards_Fr", 10, 2, 3, 2, 10)])
def test_native_fri_snark(field_name, dim, rs_extra, loc_param, interactions, queries):
    """The native prover (libiop_amd/cpp/fri.hpp behind iopx_fri_snark_prove) produces the oracle prover's transcript."""
    assert fc.native_prove_equals_oracle(emu(), torch, CPU, field_name, dim, rs_extra, loc_param, interactions, queries, 5)


def test_native_fri_snark_argument_checks():
    """iopx_fri_snark_prove: the reference's exceptions as error codes (FRI_snark_parameters: RS_extra_dimensions below the domain dimension, positive
    repetitions; more coefficients than the tested degree bound; unknown field)."""
    import numpy as np
    lib = emu()
    d = lib.malloc(64 * 24)
    try:
        lib.h2d(d, np.zeros((64, 3), dtype=np.uint64))
        with pytest.raises(ValueError):
            lib.fri_snark_prove(0, d, 64, 8, 8, 2, 1, 6)            # RS_extra_dimensions = the whole domain
        with pytest.raises(ValueError):
            lib.fri_snark_prove(0, d, 64, 8, 2, 0, 1, 6)            # localization parameter 0
        with pytest.raises(ValueError):
            lib.fri_snark_prove(0, d, 65, 8, 2, 2, 1, 6)            # 65 coefficients for a degree bound of 2^6
        with pytest.raises(ValueError):
            lib.fri_snark_prove(7, d, 64, 8, 2, 2, 1, 6)            # unknown field
        assert len(lib.fri_snark_prove(0, d, 64, 8, 2, 2, 1, 6)) > 0
        assert len(lib.fri_snark_prove(0, d, 0, 8, 2, 2, 1, 6)) > 0  # the zero polynomial
    finally:
        lib.free(d)
