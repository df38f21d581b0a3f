"""The native provers' two schedules (libiop_amd/cpp/aurora.hpp, FRI_protocol::first_round_from_head and batch_sumcheck_protocol): virtual oracles
over the HEAD of the codeword domain only (default) and over the whole domain as the reference evaluates them (IOPX_HEAD_EVAL=0) give the
oracle prover's bytes; an instance whose virtual oracles are not polynomials (an unsatisfied witness) is detected on the confirmation window
and proved by the reference's schedule, so its (rejected) transcript is still the reference's.  Kernel sources compiled for the CPU
(tests/emu); the same cases run on the MI355X in tests/test_gpu_head_eval.py."""
import ctypes

import numpy as np
import pytest
import torch

import emu_lib
import head_cases as hc
from libiop_amd import domains, r1cs

CPU = torch.device("cpu")


@pytest.mark.parametrize("protocol,field_name,log_n,num_inputs", [("aurora", "gf192", 9, 15), ("aurora", "edwards_Fr", 9, 15), ("fractal", "gf192", 7, 15),
                                                                  ("fractal", "edwards_Fr", 8, 0)])
def test_both_schedules_give_the_oracle_transcript(protocol, field_name, log_n, num_inputs, monkeypatch):
    hc.check_both_schedules(emu_lib.emu(), monkeypatch, protocol, field_name, log_n, num_inputs)


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_unsatisfied_witness_is_proved_by_the_reference_schedule(field_name, monkeypatch):
    hc.check_unsatisfied_witness(emu_lib.emu(), torch, CPU, monkeypatch, field_name)


@pytest.mark.parametrize("m,sub_dim", [(6, 2), (9, 4), (10, 7), (5, 5)])
def test_div_by_vanishing(m, sub_dim):
    hc.check_div_by_vanishing(emu_lib.emu(), torch, CPU, m, sub_dim, 40 + m)


@pytest.mark.parametrize("m,d,batch_a,batch_b,general", [(8, 5, 3, 1, False), (9, 6, 1, 1, False), (7, 6, 2, 3, False), (8, 4, 3, 1, True)])
def test_reextend_two_groups_in_one_batch(m, d, batch_a, batch_b, general):
    hc.check_reextend2(emu_lib.emu(), torch, CPU, m, d, batch_a, batch_b, 90 + m, general)


@pytest.mark.parametrize("protocol,field_name,log_n,num_inputs", [("aurora", "gf192", 8, 15), ("fractal", "edwards_Fr", 7, 0)])
def test_query_phase_behind_the_grind(protocol, field_name, log_n, num_inputs, monkeypatch):
    """The query phase enqueued behind the proof-of-work batch (IOPX_POW_BEHIND_LOG2=0: behind the very first batch, as the 2^20 proof does behind its
    2^24-candidate batch) and after the grind (the default at test sizes, whose grinds end within the short batches): the same bytes."""
    hc.check_query_phase_behind_the_grind(emu_lib.emu(), monkeypatch, protocol, field_name, log_n, num_inputs)


def test_instance_create_argument_checks():
    lib = emu_lib.emu()
    ops = domains.DeviceOps(lib, torch, CPU, domains.GF192())
    cs, primary, auxiliary = r1cs.generate_r1cs_example(ops, 64, 3, 63, 5)
    mats = [hc.csr(ops, M) for M in (cs.A, cs.B, cs.C)]
    z = np.concatenate([primary, auxiliary])
    broken = [list(m) for m in mats]
    broken[1][0] = broken[1][0].copy()
    broken[1][0][3] = broken[1][0][2] - 1 if broken[1][0][2] else 7            # offsets must not decrease
    broken[1][0][0] = 0
    with pytest.raises(ValueError):
        lib.aurora_instance(0, broken, 63, 3, z)
    wide = [list(m) for m in mats]
    wide[2][1] = wide[2][1].copy()
    wide[2][1][0] = 64                                                          # column 64 > num_variables
    with pytest.raises(ValueError):
        lib.aurora_instance(0, wide, 63, 3, z)
    with pytest.raises(ValueError):
        lib.aurora_instance(9, mats, 63, 3, z)                                  # unknown field


@pytest.mark.parametrize("count,stride,words", [(1, 1, 3), (100, 8, 3), (4096, 3, 3), (257, 16, 1)])
def test_gather_stride(count, stride, words):
    lib = emu_lib.emu()
    src = np.random.default_rng(count).integers(0, 2**63, size=(count * stride + 5, words), dtype=np.uint64)
    d_src, d_dst = lib.malloc(src.nbytes), lib.malloc(count * words * 8)
    lib.h2d(d_src, src)
    lib.c.iopx_gather_stride_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    lib._check(lib.c.iopx_gather_stride_dev(d_src, count, stride, words * 8, d_dst))
    out = np.empty((count, words), dtype=np.uint64)
    lib.d2h(out, d_dst)
    assert np.array_equal(out, src[:count * stride:stride])
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_gather_stride_dev(d_src, count, 0, words * 8, d_dst))
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_gather_stride_dev(d_src, count, stride, 12, d_dst))
    lib.free(d_src)
    lib.free(d_dst)


def test_count_mismatch():
    lib = emu_lib.emu()
    a = np.random.default_rng(3).integers(0, 2**63, size=(5000, 3), dtype=np.uint64)
    b = a.copy()
    b[17, 1] ^= np.uint64(1)
    b[4999, 2] ^= np.uint64(1 << 40)
    b[4999, 0] ^= np.uint64(2)
    d_a, d_b, d_n = lib.malloc(a.nbytes), lib.malloc(b.nbytes), lib.malloc(8)
    lib.h2d(d_a, a)
    lib.h2d(d_b, b)
    lib.h2d(d_n, np.zeros(1, dtype=np.uint64))
    lib.c.iopx_count_mismatch_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib._check(lib.c.iopx_count_mismatch_dev(d_a, d_a, a.nbytes, d_n))
    n = np.empty(1, dtype=np.uint64)
    lib.d2h(n, d_n)
    assert n[0] == 0
    lib._check(lib.c.iopx_count_mismatch_dev(d_a, d_b, a.nbytes, d_n))
    lib._check(lib.c.iopx_count_mismatch_dev(d_a, d_b, a.nbytes, d_n))        # accumulates
    lib.d2h(n, d_n)
    assert n[0] == 6
    with pytest.raises(ValueError):
        lib._check(lib.c.iopx_count_mismatch_dev(d_a, d_b, 12, d_n))
    for d in (d_a, d_b, d_n):
        lib.free(d)
