"""Generates tests/golden/reference_over_shim.json (run in the build container: needs /root/reference, libsodium, GMP).

Each entry is what the REFERENCE'S OWN prover — libiop's sources compiled unmodified by tests/harness over a stand-in libff whose field arithmetic
is this repository's host code — produced for one seeded instance: the BLAKE2b-256 digest and length of its transcript (in the byte form of
oracle::bcs_transcript::serialize), the index roots of a Fractal run, and that the reference's own verifier accepted.  The oracle and the HIP provers
are compared with these digests.  What they pin: libiop's protocol logic, run from its own code.  What they do not pin: libff's bytes (the shim is a stand-in)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
import harness  # noqa: E402


def main():
    why = harness.available()
    if why:
        sys.exit(why)
    harness.build(stubbed=False)
    entries = []
    for protocol, field, log_n, k, seed, rs_extra in harness.CASES:
        r = harness.run("plain", protocol, field, log_n, k, seed, rs_extra)
        assert r["verifier_accepts"], (protocol, field, log_n)
        entries.append({"protocol": protocol, "field": field, "log_n": log_n, "num_inputs": k, "seed": seed, "rs_extra": rs_extra, "localization": 2,
                        "reference_verifier_accepts": True, "transcript_bytes": len(r["transcript"]), "transcript_blake2b": harness.digest(r["transcript"]),
                        "index_roots": [x.hex() for x in r["index_roots"]]})
        print(protocol, field, log_n, len(r["transcript"]), entries[-1]["transcript_blake2b"][:16])
    path = os.path.join(ROOT, "tests", "golden", "reference_over_shim.json")
    keep = []
    if os.path.exists(path):
        with open(path) as f:
            keep = json.load(f).get("large_entries", [])
    with open(path, "w") as f:
        json.dump({"generated_by": "tests/golden/make_reference_over_shim.py", "program": "tests/harness/run_reference.cpp over tests/harness/shim",
                   "entries": entries, "large_entries": keep}, f, indent=1)
    functions()


def functions():
    """tests/golden/reference_functions.json: tests/harness/reference_vectors.cpp's lines (the reference's own functions on seeded inputs)."""
    import subprocess
    harness._make(["_build/reference_vectors"])
    out = subprocess.run([os.path.join(harness.HERE, "_build", "reference_vectors")], capture_output=True, text=True, check=True).stdout
    entries = [json.loads(line) for line in out.splitlines() if line.startswith("{")]
    with open(os.path.join(ROOT, "tests", "golden", "reference_functions.json"), "w") as f:
        f.write('{"generated_by": "tests/golden/make_reference_over_shim.py", "program": "tests/harness/reference_vectors.cpp over tests/harness/shim",\n "entries": [\n')
        f.write(",\n".join(json.dumps(e) for e in entries))
        f.write("\n]}\n")
    print(len(entries), "function-level vectors")


def large(specs):
    """--large aurora:gf192:16 fractal:edwards_Fr:20 ...: the reference's own prover at BASELINE's sizes (Aurora 2^20 over GF(2^192): about 70 min and 15 GB on
    one core of this container with the shim's portable field code; Fractal 2^20 over edwards_Fr: 15 min, 33 GB); entries are merged into "large_entries"."""
    import resource
    import time
    harness.build(stubbed=False)
    path = os.path.join(ROOT, "tests", "golden", "reference_over_shim.json")
    with open(path) as f:
        doc = json.load(f)
    have = {(e["protocol"], e["field"], e["log_n"]): e for e in doc.get("large_entries", [])}
    for spec in specs:
        protocol, field, log_n = spec.split(":")
        k, seed, rs_extra = (15, 0x2204, 5) if protocol == "aurora" else (15 if field == "gf192" else 0, 0x2205, 3)
        t = time.time()
        r = harness.run("plain", protocol, field, int(log_n), k, seed, rs_extra)
        assert r["verifier_accepts"], spec
        have[(protocol, field, int(log_n))] = {
            "protocol": protocol, "field": field, "log_n": int(log_n), "num_inputs": k, "seed": seed, "rs_extra": rs_extra, "localization": 2,
            "reference_verifier_accepts": True, "transcript_bytes": len(r["transcript"]), "transcript_blake2b": harness.digest(r["transcript"]),
            "index_roots": [x.hex() for x in r["index_roots"]], "seconds_on_one_core": round(time.time() - t),
            "peak_rss_mb": round(resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024)}
        print(spec, have[(protocol, field, int(log_n))]["transcript_blake2b"])
    doc["large_entries"] = [have[k] for k in sorted(have)]
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    if "--large" in sys.argv:
        large(sys.argv[sys.argv.index("--large") + 1:])
        sys.exit(0)
    if "--functions-only" in sys.argv:
        functions()
        sys.exit(0)
    main()
