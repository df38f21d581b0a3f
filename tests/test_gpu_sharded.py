"""The multi-GPU provers on the MI355X box under RCCL (`torch.distributed.run`, backend nccl; one rank per visible GPU — one on the
driver's box — and, when the box has more, every power of two up to the visible count).  One child process group per world size runs every case of the module
(tests/gpu_sharded_worker.py); the tests compare rank 0's transcript (every rank's digest is cross-checked inside the worker) with

  * oracle.aurora_prove / oracle.fractal_prove byte for byte at sizes the CPU oracle finishes in seconds, and
  * the single-GPU native prover (itself oracle-equal up to 2^14, tests/test_gpu_fullsize.py) at 2^16.

Operator sets: libiop_amd/dist.py (ShardedDeviceOps over GF(2^192): contiguous cosets; ResidueShardedDeviceOps over the 181-bit field:
residue classes) and the native ones inside the library (libiop_amd/cpp/dist.hpp; RCCL communicator created through the C ABI)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "gpu_sharded_worker.py")


def _worlds():
    import torch
    n = torch.cuda.device_count()          # counting devices does not initialise the GPU in this process
    out, w = [], 1
    while w <= max(n, 1):
        out.append(w)
        w *= 2
    return out


CASES = [("aurora", "gf192", 12, 15, 0x2204), ("fractal", "edwards_Fr", 11, 0, 0x2205), ("fractal", "gf192", 10, 15, 0x2205), ("aurora", "edwards_Fr", 11, 15, 0x2204)]
# every case of this module, run by ONE torch.distributed.run child per world size (a child per case cost ten seconds of import and group set-up each)
ALL = ([{"protocol": c[0], "field": c[1], "impl": "native", "log_n": c[2], "inputs": c[3], "seed": c[4]} for c in CASES] +
       [{"protocol": c[0], "field": c[1], "impl": "python", "log_n": c[2], "inputs": c[3], "seed": c[4]} for c in CASES[:2]] +
       [{"protocol": "aurora", "field": "gf192", "impl": "native", "log_n": 16, "inputs": 15, "seed": 0x2204},
        {"protocol": "fft", "field": "gf192", "impl": "native", "log_n": 18, "inputs": 0, "seed": 0x2201}])


def _key(c):
    return (c["protocol"], c["field"], c["impl"], c["log_n"])


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    """{world: {case key: rank 0's result}} — one child process group per world size."""
    out_dir = tmp_path_factory.mktemp("sharded")
    by_world = {}
    for world in _worlds():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = os.path.join(str(out_dir), "w%d.json" % world)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               WORKER, "--cases", json.dumps(ALL), "--out", out]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-6000:])
        with open(out) as f:
            results = json.load(f)
        assert len(results) == len(ALL)
        by_world[world] = {_key(c): res for c, res in zip(ALL, results)}
    return by_world


def _result(runs, world, protocol, field, impl, log_n):
    res = runs[world][(protocol, field, impl, log_n)]
    assert "error" not in res, res
    assert res["world"] == world and res["ranks_agree"], res
    return res


_ORACLE = {}


def _oracle(protocol, field, log_n, inputs, seed):
    """The CPU oracle's transcript (and index roots), computed once per case for both implementations."""
    key = (protocol, field, log_n, inputs, seed)
    if key not in _ORACLE:
        code = oracle.FIELD_GF192 if field == "gf192" else oracle.FIELD_EDWARDS
        _ORACLE[key] = (oracle.aurora_prove(code, log_n, inputs, seed), []) if protocol == "aurora" else oracle.fractal_prove(code, log_n, inputs, seed)
    return _ORACLE[key]


@pytest.mark.parametrize("impl,case", [("native", c) for c in CASES] + [("python", c) for c in CASES[:2]], ids=lambda v: v if isinstance(v, str) else "-".join(str(x) for x in v[:3]))
def test_sharded_prover_under_rccl_equals_the_oracle(impl, case, runs):
    protocol, field, log_n, inputs, seed = case
    ref, ref_roots = _oracle(protocol, field, log_n, inputs, seed)
    for world in _worlds():
        res = _result(runs, world, protocol, field, impl, log_n)
        assert bytes.fromhex(res["transcript"]) == ref, (impl, world, "transcript differs from the oracle prover's")
        assert [bytes.fromhex(r) for r in res["index_roots"]] == ref_roots, (impl, world)
        assert res["equals_single_gpu_native_prover"]
        if impl == "python":
            assert res["ops"] == ("ShardedDeviceOps" if field == "gf192" else "ResidueShardedDeviceOps")


def test_native_sharded_aurora_2p16_equals_the_single_gpu_prover(runs):
    for world in _worlds():
        assert _result(runs, world, "aurora", "gf192", "native", 16)["equals_single_gpu_native_prover"], world


def test_native_distributed_transform_under_rccl(runs):
    """iopx_add_fft_gf192_dist_dev / _ifft_: one 2^18-point transform block-distributed over the visible GPUs (all-to-all transpose + peer exchanges over
    RCCL; with one GPU the call is the single-GPU transform) equals the single-GPU transform, and the inverse returns the coefficients."""
    for world in _worlds():
        res = _result(runs, world, "fft", "gf192", "native", 18)
        assert res["fft_ok"] == [1] * world, (world, res)
