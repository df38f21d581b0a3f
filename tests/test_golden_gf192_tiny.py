"""The integer-only golden vectors of tests/golden/gf192_tiny.json against (a) the oracle (so the checker itself is pinned by something
that shares no code with it), (b) the product kernels compiled for the CPU, and (c) that the committed JSON is what its generator writes."""
import json
import os
import subprocess
import sys

import golden_cases as gc


def test_oracle_equals_the_integer_vectors():
    import oracle
    gc.check(oracle.additive_fft, oracle.additive_ifft, oracle.fri_fold_additive, lambda o, cs: oracle.merkle_build(o, cs, True))


def test_cpu_compiled_kernels_equal_the_integer_vectors():
    from emu_lib import emu
    lib = emu()
    gc.check(lib.additive_FFT, lib.additive_IFFT, lib.evaluate_next_f_i_over_entire_domain, lib.merkle_tree)


def test_committed_vectors_are_reproducible(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    src = open(os.path.join(here, "golden", "make_gf192_tiny.py")).read().replace('os.path.dirname(os.path.abspath(__file__))', repr(str(tmp_path)))
    script = tmp_path / "gen.py"
    script.write_text(src)
    subprocess.check_call([sys.executable, str(script)])
    assert json.load(open(tmp_path / "gf192_tiny.json")) == gc.load()
