"""Manual run (gpurun): the REFERENCE'S OWN provers — libiop's sources compiled unmodified in the build container by tests/harness, with the stubs of
INTEGRATION.md compiled in — linked against the HIP build of the library and run on the MI355X (`make -C tests/harness hip` builds
tests/harness/_hip/reference_stubbed_hip before the snapshot is sent).  Prints, per case, the reference verifier's decision, whether the transcript's digest equals
the committed one (tests/golden/reference_over_shim.json, oracle_*_transcript_digests_large.json), the kernels the reference's prover launched and the
wall-clock of the whole program (generator, prover, verifier: the reference's host code around the stubs is unchanged and single-threaded)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "harness", "_hip", "reference_stubbed_hip")


def expected():
    out = {}
    with open(os.path.join(ROOT, "tests", "golden", "reference_over_shim.json")) as f:
        for e in json.load(f)["entries"]:
            out[(e["protocol"], e["field"], e["log_n"])] = e["transcript_blake2b"]
    with open(os.path.join(ROOT, "tests", "golden", "oracle_aurora_transcript_digests_large.json")) as f:
        for k, v in json.load(f)["digests"].items():
            out.setdefault(("aurora", "gf192", int(k)), v["transcript_blake2b"])
    with open(os.path.join(ROOT, "tests", "golden", "oracle_fractal_transcript_digests_large.json")) as f:
        for k, v in json.load(f)["digests"].items():
            out.setdefault(("fractal", "edwards_Fr", int(k)), v["transcript_blake2b"])
    return out


def main():
    want = expected()
    if "--headline" in sys.argv:            # BASELINE's headline instance through the reference's own prover: minutes of the reference's host code around the kernels
        cases = [("aurora", "gf192", 18, 15, 0x2204, 5), ("aurora", "gf192", 20, 15, 0x2204, 5), ("fractal", "edwards_Fr", 20, 0, 0x2205, 3)]
    else:
        cases = [("aurora", "gf192", 12, 15, 0x2204, 5), ("aurora", "gf192", 14, 15, 0x2204, 5), ("aurora", "gf192", 16, 15, 0x2204, 5),
                 ("aurora", "edwards_Fr", 12, 15, 0x2204, 5), ("fractal", "gf192", 9, 15, 0x2205, 3), ("fractal", "edwards_Fr", 16, 0, 0x2205, 3),
                 ("ligero", "gf192", 10, 15, 0x2206, 2)]
    if "--aurora-2p21" in sys.argv:        # beyond every committed digest: compared by hand with bench.py --log-n 21's config.transcript_blake2b
        cases = [("aurora", "gf192", 21, 15, 0x2204, 5)]
    if "--fractal-only" in sys.argv:
        cases = [c for c in cases if c[0] == "fractal"]
    ok = True
    for proto, field, log_n, k, seed, rs in cases:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "t.bin")
            t = time.time()
            r = subprocess.run([EXE, proto, field, str(log_n), str(k), hex(seed), str(rs), "2", out], capture_output=True, text=True, timeout=5000)
            dt = time.time() - t
            if r.returncode not in (0, 1):
                print(json.dumps({"case": [proto, field, log_n], "error": r.stderr[-400:]}))
                ok = False
                continue
            info = json.loads(r.stdout.strip().splitlines()[-1])
            with open(out, "rb") as f:
                digest = hashlib.blake2b(f.read(), digest_size=32).hexdigest()
        exp = want.get((proto, field, log_n))
        line = {"case": [proto, field, log_n], "reference_verifier_accepts": info["verifier_accepts"], "transcript_blake2b": digest,
                "equals_committed_digest": (digest == exp) if exp else None, "program_seconds": round(dt, 2),
                "kernel_launches_in_prover": info["kernel_launches_in_prover"]}
        ok = ok and info["verifier_accepts"] and (exp is None or digest == exp)
        print(json.dumps(line))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
