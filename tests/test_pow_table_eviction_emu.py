"""the two-level power-table cache of the prime-field kernels evicts while a host call still holds borrowed
tables.  With the cap forced to 2 entries every build_two_level() of a multi-table call (sumcheck g: zhi/zlo + ihi/ilo; the LDT
combination: one pair per degree gap; the coset FFT) evicts tables borrowed earlier in the same call — the results must still equal
the oracle's (the borrowed tables are kept alive by shared ownership).  The cap is read once per process, hence the subprocess."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_results_survive_eviction_of_borrowed_tables():
    env = dict(os.environ, IOPX_POW_TABLE_CAP="2")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
           "tests/test_ldt_emu.py::test_multiplicative", "tests/test_ldt_emu.py::test_sumcheck_g_multiplicative",
           "tests/test_ldt_emu.py::test_fz_multiplicative", "tests/test_ldt_emu.py::test_rowcheck_multiplicative",
           "tests/test_fractal_emu.py::test_domain_kernels"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
