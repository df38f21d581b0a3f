"""Build libiop_amd/lib/libiop_amd.so from the HIP sources (hipcc, gfx950 only)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libiop_amd.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
               "-fno-gpu-rdc", "-I" + os.path.join(CSRC, "include"),
               # `#pragma unroll` is a directive here, not a hint: the prime-field kernels keep whole radix-8 butterfly groups
               # and Poseidon states in registers, which needs every limb loop flattened (else the arrays land in scratch)
               "-mllvm", "-pragma-unroll-threshold=1000000"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs] + [os.path.join(_HERE, "..", "include", "libiop_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + sources() + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
