"""Kernel branches that a process-wide tuning value selects, on the MI355X: each case re-runs transform shapes of tests/test_gpu_parity.py
against the oracle in a CHILD process with the environment that takes the branch (the values are read once per process).

  IOPX_RS_COMB_CAP_LOG2=3   per-byte shift-term tables (k_rs_tables) + small numerators over many coset bits: what the prover's f_1v
                            (16 coefficients -> 2^25 points, r1cs_rs_iop.tcc:213) runs with the default cap, here at test sizes too
  IOPX_P1_COMB=0            phase-1 twists on the general product only
  IOPX_EDGE_BATCH=0         every polynomial of a batch takes its own last pass
  IOPX_SMALL_LAST=0         general product at the last two levels (no one- / two-word numerators)
  IOPX_COMB=0               general product everywhere (fft.tcc:39-124 has one multiplier; every branch must agree with it)
  IOPX_EDGE_MULTI=0 / 3     the single-polynomial edge passes one coset at a time (k_bfly_edge), or three cosets of a tile position per workgroup
                            (k_bfly_edge_multi, default four), also with the byte tables of shift terms
  IOPX_DEFER_ROOTS=0        (provers) every Merkle root read back at its round end instead of with the query phase's read-backs
  IOPX_MERKLE_STREAM=0      (provers) no side stream: every round's Merkle tree on the main stream

and, with the default environment, the exact f_1v shape (16 coefficients over the 2^25-point codeword domain) sampled against
oracle.poly_eval."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = r"""
import numpy as np
import oracle
import libiop_amd
from helpers import rand_elems, one_word_basis
W = 3
lib = libiop_amd.lib()
lib.init(0)
def dom(m, kind, seed):
    if kind == "aurora":
        return oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
    if kind == "std0":
        return oracle.standard_basis(m, W), np.zeros(W, dtype=np.uint64)
    return rand_elems(seed + 1, m, W), rand_elems(seed, 1, W)[0]
# test_fft_lde's shapes plus very short polynomials (many cosets), standard and general bases
for m, ncoef in [(4, 1), (6, 5), (10, 255), (12, 16), (13, 5), (13, 300), (14, 16), (15, 4096), (16, 2), (17, 8195), (18, 16)]:
    for kind in ("aurora", "general"):
        if kind == "general" and m > 15:
            continue
        basis, shift = dom(m, kind, 7 + m)
        coeffs = rand_elems(m + ncoef, ncoef, W)
        assert np.array_equal(lib.additive_FFT(coeffs, basis, shift), oracle.additive_fft(coeffs, basis, shift)), (m, ncoef, kind)
# test_fft_full_size / test_ifft shapes
for m in (1, 3, 6, 11, 12, 13, 16):
    for kind in ("std0", "aurora", "general"):
        basis, shift = dom(m, kind, 100 + m)
        coeffs = rand_elems(m, 1 << m, W)
        evals = oracle.additive_fft(coeffs, basis, shift)
        assert np.array_equal(lib.additive_FFT(coeffs, basis, shift), evals), (m, kind)
        assert np.array_equal(lib.additive_IFFT(evals, basis, shift), coeffs), (m, kind)
# test_one_word_last_levels' shapes
for m, k, shift0, second in [(2, 0, 5, False), (3, 31, 0xFFFFFFFF, False), (11, 1, 0x80000001, False), (13, 7, 1 << 20, False), (16, 20, 0xABCDEF01, False),
                             (3, 1, 3, True), (4, 31, 0xFFFFFFFF, True), (12, 30, 1, True), (16, 18, 0xFFFF0000, True)]:
    basis = one_word_basis(m, k, 900 + m, second)
    shift = np.array([shift0, 0, 0], dtype=np.uint64)
    coeffs = rand_elems(70 + m, 1 << m, W)
    evals = oracle.additive_fft(coeffs, basis, shift)
    assert np.array_equal(lib.additive_FFT(coeffs, basis, shift), evals)
    assert np.array_equal(lib.additive_IFFT(evals, basis, shift), coeffs)
    if m >= 4:
        short = rand_elems(71 + m, 1 << (m - 2), W)
        assert np.array_equal(lib.additive_FFT(short, basis, shift), oracle.additive_fft(short, basis, shift))
        tiny = rand_elems(72 + m, 3, W)
        assert np.array_equal(lib.additive_FFT(tiny, basis, shift), oracle.additive_fft(tiny, basis, shift))
# batched low-degree extensions and the re-extension path (the shared last pass of a batch)
import torch
dev = torch.device("cuda:0")
lib.set_stream(torch.cuda.current_stream().cuda_stream)
from libiop_amd import domains
ops = domains.DeviceOps(lib, torch, dev, domains.GF192())
for m, d, batch in [(16, 11, 3), (17, 12, 4), (15, 11, 2), (14, 4, 3)]:
    basis, shift = dom(m, "aurora", 3)
    D = domains.Domain(domains.GF192(), domains.ADDITIVE, basis=basis, shift=shift)
    polys = [rand_elems(40 + m + q, 1 << d, W) for q in range(batch)]
    outs = ops.FFT_batch([ops.upload(p) for p in polys], 1 << d, D)
    for p, o in zip(polys, outs):
        assert np.array_equal(ops.download(o), oracle.additive_fft(p, basis, shift)), (m, d, batch)
    H = D.get_subset_of_order(1 << d)
    evs = [oracle.additive_fft(p, H.basis, H.shift) for p in polys]
    packed = ops.upload(np.concatenate(evs))
    outs = ops.reextend_packed(packed, batch, H, D)
    for p, o in zip(polys, outs):
        assert np.array_equal(ops.download(o), oracle.additive_fft(p, basis, shift)), ("reextend", m, d, batch)
print("ok")
"""

F1V = r"""
import numpy as np
import oracle
import libiop_amd
from helpers import rand_elems
import torch
W, m = 3, 25
lib = libiop_amd.lib()
lib.init(0)
lib.set_stream(torch.cuda.current_stream().cuda_stream)
from libiop_amd import domains
ops = domains.DeviceOps(lib, torch, torch.device("cuda:0"), domains.GF192())
basis, shift = oracle.standard_basis(m, W), np.array([1 << m, 0, 0], dtype=np.uint64)
D = domains.Domain(domains.GF192(), domains.ADDITIVE, basis=basis, shift=shift)
coeffs = rand_elems(0x1f1, 16, W)
out = ops.FFT(ops.upload(coeffs), 16, D)
rng = np.random.Generator(np.random.PCG64(5))
pos = sorted(set([0, 1, 15, 16, 17, 255, 256, (1 << 21) - 1, 1 << 21, (1 << 24) - 1, 1 << 24, (1 << 25) - 1] + [int(v) for v in rng.integers(0, 1 << m, size=40)]))
# standard basis: element i of the domain is the field element with integer representation i ^ 2^25 (subspace.tcc:56-71)
def point(i):
    return np.array([i ^ (1 << m), 0, 0], dtype=np.uint64)
idx = torch.tensor(pos, dtype=torch.int64, device="cuda:0")
got = out[idx].cpu().numpy().view(np.uint64)
for row, p in enumerate(pos):
    assert np.array_equal(got[row], oracle.poly_eval(coeffs, point(p))), p
# whole cosets of the 16-point transform: the first, the last and one in the middle
for lo in (0, (1 << m) - 16, 12345 * 16):
    blk = out[lo:lo + 16].cpu().numpy().view(np.uint64)
    for i in range(16):
        assert np.array_equal(blk[i], oracle.poly_eval(coeffs, point(lo + i))), (lo, i)
print("ok")
"""


def _run(script, extra_env, timeout=1500):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]), **extra_env)
    out = subprocess.run([sys.executable, "-c", script], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (extra_env, out.stdout[-2000:], out.stderr[-4000:])


# one child process per COMBINATION (a child costs ten seconds of the driver's suite): every option appears with its non-default value at least once,
# and the options that act on the same kernel appear separately
@pytest.mark.parametrize("env", [{"IOPX_RS_COMB_CAP_LOG2": "3"}, {"IOPX_P1_COMB": "0", "IOPX_EDGE_BATCH": "0", "IOPX_EDGE_MULTI": "0", "IOPX_SMALL_LAST": "0"},
                                 {"IOPX_COMB": "0"}, {"IOPX_EDGE_MULTI": "3", "IOPX_RS_COMB_CAP_LOG2": "3"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_env_gated_branches_equal_the_oracle(env):
    _run(SHAPES, env)


PROVERS = r"""
import oracle
import libiop_amd
lib = libiop_amd.lib()
lib.init(0)
for field, code, log_n in ((0, oracle.FIELD_GF192, 11), (1, oracle.FIELD_EDWARDS, 12)):
    n = 1 << log_n
    inst = lib.aurora_example_instance(field, n, 15, n - 1, 0x2204)
    assert lib.aurora_prove(inst) == oracle.aurora_prove(code, log_n, 15, 0x2204)
    lib.aurora_instance_free(inst)
inst = lib.aurora_example_instance(1, 1024, 0, 1023, 0x2205)
ref, ref_roots = oracle.fractal_prove(oracle.FIELD_EDWARDS, 10, 0, 0x2205)
assert lib.fractal_index(inst) == ref_roots and lib.fractal_prove(inst) == ref and lib.fractal_prove(inst) == ref
lib.aurora_instance_free(inst)
print("ok")
"""


@pytest.mark.parametrize("env", [{"IOPX_DEFER_ROOTS": "0"}, {"IOPX_MERKLE_STREAM": "0"}], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_provers_with_roots_read_at_round_ends_or_with_the_queries(env):
    _run(PROVERS, env)


def test_f1v_shape_16_coefficients_over_2p25_points():
    """The transform of the prover's f_1v at BASELINE's size, default tuning (2^21 cosets x 4 > the 2^22-entry cap: byte tables)."""
    _run(F1V, {})
