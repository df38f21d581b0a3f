"""The native provers' two schedules on the MI355X (tests/head_cases.py; the CPU-compiled run of the same cases is tests/test_head_eval_emu.py):
virtual oracles over the head of the codeword domain (default) and over the whole domain (IOPX_HEAD_EVAL=0) give the oracle prover's bytes, and an
unsatisfied witness is detected on the confirmation window and proved by the reference's schedule."""
import pytest
import torch

import head_cases as hc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import libiop_amd
    lib = libiop_amd.lib()
    lib.init(0)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)      # DeviceOps (the second prover) shares torch's stream
    return lib


@pytest.mark.parametrize("protocol,field_name,log_n,num_inputs", [("aurora", "gf192", 12, 15), ("aurora", "edwards_Fr", 11, 15), ("fractal", "gf192", 10, 15),
                                                                  ("fractal", "edwards_Fr", 11, 0)])
def test_both_schedules_give_the_oracle_transcript(gpu, protocol, field_name, log_n, num_inputs, monkeypatch):
    hc.check_both_schedules(gpu, monkeypatch, protocol, field_name, log_n, num_inputs)


@pytest.mark.parametrize("field_name", ["gf192", "edwards_Fr"])
def test_unsatisfied_witness_is_proved_by_the_reference_schedule(gpu, field_name, monkeypatch):
    hc.check_unsatisfied_witness(gpu, torch, torch.device("cuda:0"), monkeypatch, field_name, log_n=10)


@pytest.mark.parametrize("m,sub_dim", [(10, 4), (16, 4), (14, 9)])
def test_div_by_vanishing(gpu, m, sub_dim):
    hc.check_div_by_vanishing(gpu, torch, torch.device("cuda:0"), m, sub_dim, 60 + m)


@pytest.mark.parametrize("m,d,batch_a,batch_b,general", [(16, 11, 3, 1, False), (17, 12, 3, 1, False), (13, 10, 2, 3, False), (14, 11, 3, 1, True)])
def test_reextend_two_groups_in_one_batch(gpu, m, d, batch_a, batch_b, general):
    hc.check_reextend2(gpu, torch, torch.device("cuda:0"), m, d, batch_a, batch_b, 70 + m, general)


@pytest.mark.parametrize("protocol,field_name,log_n,num_inputs", [("aurora", "gf192", 12, 15), ("fractal", "edwards_Fr", 10, 0)])
def test_query_phase_behind_the_grind(gpu, protocol, field_name, log_n, num_inputs, monkeypatch):
    hc.check_query_phase_behind_the_grind(gpu, monkeypatch, protocol, field_name, log_n, num_inputs)
