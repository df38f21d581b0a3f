#!/bin/bash
# A/B of builds of the library on the Aurora 2^20 bench (one box, same session): tools/ab_lib.sh OUT ROUNDS libA.so libB.so ...
# (copies each over libiop_amd/lib/libiop_amd.so in turn; on the GPU box's scratch copy of the repo only)
out=$1; n=$2; shift 2
: > "$out"
for i in $(seq $n); do
  for l in "$@"; do
    cp "$l" libiop_amd/lib/libiop_amd.so
    bash tools/ab_bench.sh /tmp/ab_one.txt "IOPX_AB_LIB=$(basename $l)" > /dev/null
    cat /tmp/ab_one.txt >> "$out"
  done
done
cat "$out"
