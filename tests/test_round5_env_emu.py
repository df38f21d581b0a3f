"""Options of the library on the CPU-compiled kernels, each in a child process that finds them in its environment (the library asks the environment once
per name):
  IOPX_DEFER_ROOTS=0   every Merkle root read back at its round end (round 4's schedule) instead of with the query phase's read-backs
  IOPX_MERKLE_STREAM=0 no side stream: every round's Merkle tree on the main stream
  IOPX_EDGE_MULTI=0/1/3 the single-polynomial edge passes one coset at a time (k_bfly_edge), or 1 / 3 cosets of a tile position per workgroup
                       (k_bfly_edge_multi; default 4); with IOPX_RS_COMB_CAP_LOG2=0 the shift terms come from the byte tables
The provers must give the oracle's bytes and the transforms the oracle's values on every branch."""
import os
import subprocess
import sys

import pytest

PROVERS = r"""
import oracle
from emu_lib import emu
lib = emu()
for field, code in ((0, oracle.FIELD_GF192), (1, oracle.FIELD_EDWARDS)):
    inst = lib.aurora_example_instance(field, 256, 15, 255, 0x2204)
    assert lib.aurora_prove(inst) == oracle.aurora_prove(code, 8, 15, 0x2204)
    lib.aurora_instance_free(inst)
    inst = lib.aurora_example_instance(field, 128, 0, 127, 0x2205)
    ref, ref_roots = oracle.fractal_prove(code, 7, 0, 0x2205)
    assert lib.fractal_index(inst) == ref_roots and lib.fractal_prove(inst) == ref and lib.fractal_prove(inst) == ref
    lib.aurora_instance_free(inst)
print("ok")
"""

TRANSFORMS = r"""
import numpy as np
import torch
import oracle
from emu_lib import emu
from helpers import rand_elems
from libiop_amd import domains
W = 3
lib = emu()
ops = domains.DeviceOps(lib, torch, torch.device("cpu"), domains.GF192())
for m, kind in ((12, "std"), (13, "general"), (11, "std")):
    basis = oracle.standard_basis(m, W) if kind == "std" else rand_elems(50 + m, m, W)
    shift = np.array([1 << m, 0, 0], dtype=np.uint64) if kind == "std" else rand_elems(51 + m, 1, W)[0]
    coeffs = rand_elems(20 + m, 1 << m, W)
    ev = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(lib.additive_FFT(coeffs, basis, shift), ev) and np.array_equal(lib.additive_IFFT(ev, basis, shift), coeffs), (m, kind)
# the batched last pass (2 - 4 polynomials of one coset range)
for m, d, batch in ((14, 11, 4), (14, 11, 3), (13, 11, 2)):
    basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
    D = domains.Domain(domains.GF192(), domains.ADDITIVE, basis=basis, shift=shift)
    polys = [rand_elems(40 + m + q, 1 << d, W) for q in range(batch)]
    for p, o in zip(polys, ops.FFT_batch([ops.upload(p) for p in polys], 1 << d, D)):
        assert np.array_equal(ops.download(o), oracle.additive_fft(p, basis, shift)), (m, d, batch)
print("ok")
"""

# extensions of one polynomial over several cosets (standard and general bases), and the batched inverse: the shapes k_bfly_edge_multi takes
EXTENSIONS = r"""
import numpy as np
import torch
import oracle
from emu_lib import emu
from helpers import rand_elems
from libiop_amd import domains
W = 3
lib = emu()
ops = domains.DeviceOps(lib, torch, torch.device("cpu"), domains.GF192())
for m, d, kind in ((14, 11, "std"), (13, 10, "general"), (15, 12, "std"), (12, 11, "general")):
    basis = oracle.standard_basis(m, W) if kind == "std" else rand_elems(60 + m, m, W)
    shift = np.array([1 << m, 0, 0], dtype=np.uint64) if kind == "std" else rand_elems(61 + m, 1, W)[0]
    D = domains.Domain(domains.GF192(), domains.ADDITIVE, basis=basis, shift=shift)
    poly = rand_elems(70 + m, 1 << d, W)
    ev = oracle.additive_fft(poly, basis, shift)
    assert np.array_equal(ops.download(ops.FFT(ops.upload(poly), 1 << d, D)), ev), (m, d, kind)
for m, batch, kind in ((11, 3, "std"), (12, 5, "general")):
    basis = oracle.standard_basis(m, W) if kind == "std" else rand_elems(80 + m, m, W)
    shift = np.array([1 << m, 0, 0], dtype=np.uint64) if kind == "std" else rand_elems(81 + m, 1, W)[0]
    D = domains.Domain(domains.GF192(), domains.ADDITIVE, basis=basis, shift=shift)
    polys = [rand_elems(90 + m + q, 1 << m, W) for q in range(batch)]
    evs = [oracle.additive_fft(p, basis, shift) for p in polys]
    for p, o in zip(polys, ops.IFFT_batch([ops.upload(e) for e in evs], D)):
        assert np.array_equal(ops.download(o), p), (m, batch, kind)
print("ok")
"""


def _run(script, extra_env):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests")]), **extra_env)
    out = subprocess.run([sys.executable, "-c", script], env=env, cwd=root, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (extra_env, out.stdout[-2000:] + out.stderr[-4000:])


def test_roots_read_at_every_round_end():
    _run(PROVERS, {"IOPX_DEFER_ROOTS": "0"})


def test_trees_on_the_main_stream():
    _run(PROVERS, {"IOPX_MERKLE_STREAM": "0"})


@pytest.mark.parametrize("env", [{"IOPX_EDGE_MULTI": "0"}, {"IOPX_EDGE_MULTI": "1"}, {"IOPX_EDGE_MULTI": "3"}, {},
                                 {"IOPX_EDGE_MULTI": "3", "IOPX_RS_COMB_CAP_LOG2": "0"}, {"IOPX_SMALL_LAST": "0"}],
                         ids=["one-coset-kernel", "multi-1", "multi-3", "default", "multi-3-byte-tables", "no-small-numerators"])
def test_edge_pass_cosets_per_workgroup(env):
    _run(EXTENSIONS, env)
    _run(TRANSFORMS, env)
