for cfg in "" "IOPX_EDGE_THREADS=128" "IOPX_EDGE_THREADS=512" "IOPX_P2_TOP=3" "IOPX_P2_TOP=5" "IOPX_EDGE_TILE_BITS=9" "IOPX_EDGE_TILE_BITS=11" "IOPX_EDGE_TILE_BITS=11 IOPX_EDGE_THREADS=512" "IOPX_P2_TOP=2" "IOPX_P2_THREADS=256" "IOPX_TILE_BITS=12" "IOPX_TILE_BITS=10"; do
  echo "== $cfg"; env $cfg python tools/lde_bench.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); l=d['lde_2^20->2^25']; print({k:l[k] for k in ('total_ms','k_bfly_upper','k_bfly_edge','k_phase1') if k in l}, 'fft22', d['fft_2^22']['total_ms'])"
done
