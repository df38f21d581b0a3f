"""TEST INFRASTRUCTURE ONLY: builds and loads tests/emu/libiopx_emu.so — the product kernel sources compiled
for the CPU with one thread per workgroup (see tests/emu/fakehip/hip/hip_runtime.h).  Wrapped by the
same ctypes binding class as the product library so the tests call the C ABI exactly as on the GPU."""
import os
import subprocess

import libiop_amd

_EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
_emu = None


def emu():
    global _emu
    if _emu is None:
        override = os.environ.get("IOPX_EMU_LIB")       # e.g. an AddressSanitizer build of the same sources (CPU only)
        if override:
            _emu = libiop_amd.Library(override)
        else:
            import fcntl
            with open(os.path.join(_EMU_DIR, ".build.lock"), "w") as lock:       # ranks and xdist workers start together: one build at a time
                fcntl.flock(lock, fcntl.LOCK_EX)
                subprocess.check_call(["make", "-s", "-j8", "-C", _EMU_DIR])
            _emu = libiop_amd.Library(os.path.join(_EMU_DIR, "libiopx_emu.so"))
    return _emu
