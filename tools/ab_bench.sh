#!/bin/bash
# A/B of tuning environment variables on the Aurora 2^20 bench (one box, same session): tools/ab_bench.sh OUT "ENV1" "ENV2" ...
out=$1; shift
: > "$out"
for e in "$@"; do
  echo "== $e" >> "$out"
  env $e python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j['roofline']
k = r['kernels_ms_per_step']; pre = lambda p: sum(v for n, v in k.items() if n.startswith(p)); print('ms_per_step %.3f  upper %.3f  edge %.3f (batch %.3f fwd %.3f inv %.3f)  merkle %.3f  kernels %.3f  host_gap %.3f  alu_frac %s  digest %s' % (j['ms_per_step'], pre('k_bfly_upper'), pre('k_bfly_edge'), k.get('k_bfly_edge_fwd_batch', 0), k.get('k_bfly_edge_fwd', 0), k.get('k_bfly_edge_inv', 0), pre('k_merkle'), r['kernels_ms_total'], r['host_gap_ms'], r.get('alu_ceiling_frac'), j['config'].get('transcript_blake2b', '')[:16]))
" >> "$out"
done
cat "$out"
