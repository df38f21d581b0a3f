"""Multi-pass tile schedules of the FFT kernels, exercised on the CPU emulation with small tiles."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("tile,p1c,p2c,p2t,ms", [
    (5, 2, 2, 2, "4,5,6,7,9,11,12"),
    (6, 1, 3, 3, "6,7,10,13"),
    (4, 1, 1, 1, "5,8,10"),
    (7, 2, 4, 4, "7,8,12,14"),
    (8, 0, 0, 0, "9,12"),
])
def test_schedules(tile, p1c, p2c, p2t, ms):
    env = dict(os.environ, IOPX_TILE_BITS=str(tile), IOPX_P1_COLS=str(p1c), IOPX_P2_COLS=str(p2c), IOPX_P2_TOP=str(p2t))
    r = subprocess.run([sys.executable, os.path.join(HERE, "emu_schedule_check.py"), ms], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
