"""Build libiop_amd/lib/libiop_amd.so from the HIP sources (hipcc, gfx950 only): one object per source, compiled in parallel, then linked."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
CPP = os.path.join(_HERE, "cpp")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libiop_amd.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-fno-gpu-rdc", "-I" + os.path.join(CSRC, "include"),
               # `#pragma unroll` is a directive here, not a hint: the prime-field kernels keep whole radix-8 butterfly groups
               # and Poseidon states in registers, which needs every limb loop flattened (else the arrays land in scratch)
               "-mllvm", "-pragma-unroll-threshold=1000000"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    """Everything a source may include: the kernel headers, the public C header and the C++ prover headers of libiop_amd/cpp
    (prover_capi.hip compiles those into the library)."""
    hs = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs if not f.endswith(".hip")]
    hs += [os.path.join(CPP, f) for f in os.listdir(CPP)]
    hs.append(os.path.join(_HERE, "..", "include", "libiop_amd.h"))
    return hs


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stale_objects():
    newest_header = max(os.path.getmtime(h) for h in _headers())
    out = []
    for src in sources():
        o = _obj(src)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), newest_header):
            out.append(src)
    return out


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = sources() if force else _stale_objects()

    def compile_one(src):
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + [_obj(s) for s in sources()] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
