"""TEST INFRASTRUCTURE ONLY — container-only: build and run the reference-over-shim programs of tests/harness.

`reference_plain`    the reference's own Aurora / Fractal prover and verifier, compiled straight from /root/reference over the stand-in libff /
                     libfqfft of tests/harness/shim (field arithmetic = this repository's host code);
`reference_stubbed`  the same program with the stubs of INTEGRATION.md compiled in verbatim (tests/harness/make_shadow.py), forwarding to the CPU build
                     of the kernel sources (tests/emu) through the C ABI.
Nothing of the reference is copied; the binaries live in tests/harness/_build (git-ignored, gpurun-ignored)."""
import hashlib
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
CONDA = "/opt/conda"

# (protocol, field, log_n, num_inputs, seed, rs_extra): the instrument programs' settings (instrument_aurora_snark.cpp:95-122: RS_extra_dimensions 5 non-zk,
# k = 15; instrument_fractal_snark.cpp:93-110: 3, k = 15 over subspaces and 0 over cosets), localization parameter 2
CASES = [("aurora", "gf192", 6, 15, 0x2204, 5), ("aurora", "gf192", 8, 15, 0x2204, 5), ("aurora", "gf192", 10, 15, 0x2204, 5), ("aurora", "gf192", 12, 15, 0x2204, 5),
         ("aurora", "edwards_Fr", 8, 15, 0x2204, 5), ("aurora", "edwards_Fr", 10, 15, 0x2204, 5), ("aurora", "edwards_Fr", 12, 15, 0x2204, 5),
         ("fractal", "gf192", 7, 15, 0x2205, 3), ("fractal", "gf192", 9, 15, 0x2205, 3),
         ("fractal", "edwards_Fr", 8, 0, 0x2205, 3), ("fractal", "edwards_Fr", 11, 0, 0x2205, 3)]


def available():
    """None when the harness can be built here, else the reason it cannot."""
    if not os.path.isdir(os.path.join(REFERENCE, "libiop")):
        return "no /root/reference (the reference tree exists in the build container only)"
    for f in ("include/sodium.h", "include/gmp.h", "lib/libsodium.so.23", "lib/libgmp.so.10"):
        if not os.path.exists(os.path.join(CONDA, f)):
            return "no %s/%s (libiop needs libsodium and GMP)" % (CONDA, f)
    return None


def _make(targets):
    """xdist workers start together: one make at a time in this directory"""
    import fcntl
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-j4", "-C", HERE] + list(targets))


def build(stubbed=True):
    _make(["_build/reference_plain"] + (["_build/reference_stubbed"] if stubbed else []))


def run(variant, protocol, field, log_n, num_inputs, seed, rs_extra, localization=2):
    """-> {"transcript": bytes, "verifier_accepts": bool, "index_roots": [bytes], "kernel_launches_in_prover": {name: count}}"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "t.bin")
        r = subprocess.run([os.path.join(HERE, "_build", "reference_" + variant), protocol, field, str(log_n), str(num_inputs), hex(seed), str(rs_extra),
                            str(localization), out], capture_output=True, text=True, timeout=1800)
        assert r.returncode in (0, 1), (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        with open(out, "rb") as f:
            info["transcript"] = f.read()
    info["index_roots"] = [bytes.fromhex(x) for x in info["index_roots"]]
    return info


def digest(b):
    return hashlib.blake2b(b, digest_size=32).hexdigest()


# libiop's own test files run as programs (tests/harness/Makefile REFTESTS); the default subset (three programs, seconds each) keeps the CPU suite short, IOPX_REFTESTS=all runs the 34
REFTESTS_DEFAULT = ["protocols/test_fri_aux", "protocols/test_aurora_protocol", "protocols/test_direct_ldt"]


def reftests():
    if os.environ.get("IOPX_REFTESTS") == "all":
        out = subprocess.run(["make", "-s", "-C", HERE, "-pn"], capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("REFTESTS :="):
                return line.split(":=")[1].split()
    return list(REFTESTS_DEFAULT)


def build_reftest(name, variant="stubbed"):
    _make(["_build/reftests/%s/%s" % (variant, name)])


# the two tests that feed raw random bytes as edwards_Fr elements (tests/harness/run_reftests.py FILTER)
REFTEST_FILTER = {"protocols/test_ligero_interleaved_lincheck_et": "--gtest_filter=-InterleavedLincheckETTrueMultiplicativeTest.*",
                  "protocols/test_ligero_interleaved_lincheck_ot": "--gtest_filter=-InterleavedLincheckOTTrueMultiplicativeTest.*"}


def run_reftest(name, variant="stubbed"):
    """-> (tests run, tests passed, {kernel: launches}, tail of the output)"""
    r = subprocess.run([os.path.join(HERE, "_build", "reftests", variant, name)] + ([REFTEST_FILTER[name]] if name in REFTEST_FILTER and variant != "plain" else []),
                       capture_output=True, text=True, timeout=1800)
    lines = r.stdout.splitlines()
    ran = sum(1 for l in lines if l.startswith("[ RUN "))
    ok = sum(1 for l in lines if l.startswith("[       OK ]"))
    kernels = {}
    if "[ KERNELS  ]" in r.stdout:
        for l in r.stdout.split("[ KERNELS  ]")[1].split("[ /KERNELS ]")[0].splitlines():
            parts = l.split()
            if len(parts) >= 2 and parts[0].startswith("k_"):
                kernels[parts[0]] = int(parts[1])
    return ran, ok, kernels, (r.stdout[-600:] + r.stderr[-600:]) if r.returncode else ""
