#!/bin/bash
# A/B of builds of the library on the native Fractal 2^20 prover (one box, same session): tools/ab_lib_fractal.sh OUT ROUNDS libA.so libB.so ...
out=$1; n=$2; shift 2
: > "$out"
for i in $(seq $n); do
  for l in "$@"; do
    cp "$l" libiop_amd/lib/libiop_amd.so
    echo "== $(basename $l)" >> "$out"
    python3 tools/fractal_bench.py --log-n 20 --reps 3 --native --native-only --out /tmp/fr_one.json > /dev/null 2>&1
    python3 - >> "$out" <<'PY'
import json
j = json.load(open("/tmp/fr_one.json"))
n = j["native"]
print("prover_ms_min %.3f  k_mfft_pass %.3f ms  kernels %.3f ms" % (1e3 * n["prover_s_min"], n["kernels"].get("k_mfft_pass", {}).get("ms", 0), n["kernels_ms_total"]))
PY
  done
done
cat "$out"
