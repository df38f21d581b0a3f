"""TEST INFRASTRUCTURE ONLY: run the reference's own test programs built by `make reftests_plain` / `reftests_stubbed`; one summary line per program."""
import glob, os, subprocess, sys, time, json
HERE = os.path.dirname(os.path.abspath(__file__))
variant = sys.argv[1]
out = {}
for exe in sorted(glob.glob(os.path.join(HERE, "_build", "reftests", variant, "*", "*"))):
    name = os.path.relpath(exe, os.path.join(HERE, "_build", "reftests", variant))
    t = time.time()
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=float(os.environ.get("REFTEST_TIMEOUT", "1500")))
        lines = r.stdout.splitlines()
        ran = [l for l in lines if l.startswith("[ RUN ")]
        ok = [l for l in lines if l.startswith("[       OK ]")]
        failed = [l.split("]")[1].strip() for l in lines if l.startswith("[  FAILED  ]") and "." in l]
        kernels = {}
        if "[ KERNELS  ]" in r.stdout:
            for l in r.stdout.split("[ KERNELS  ]")[1].split("[ /KERNELS ]")[0].splitlines():
                parts = l.split()
                if len(parts) >= 2 and parts[0].startswith("k_"): kernels[parts[0]] = int(parts[1])
        out[name] = {"rc": r.returncode, "tests": len(ran), "passed": len(ok), "failed": failed, "seconds": round(time.time() - t, 1), "kernel_launches": kernels}
        if r.returncode not in (0, 1): out[name]["stderr"] = r.stderr[-300:]
    except subprocess.TimeoutExpired:
        out[name] = {"rc": "timeout", "seconds": round(time.time() - t, 1)}
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(HERE, "_build", "reftests_%s.json" % variant), "w"), indent=1)
print("programs", len(out), "tests", sum(v.get("tests", 0) for v in out.values()), "passed", sum(v.get("passed", 0) for v in out.values()))
