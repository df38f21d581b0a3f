#!/bin/bash
# Profile collection on the GPU box (run from the repo root through gpurun): for the Aurora bench command and for the Fractal prover
# (tools/fractal_bench.py, BASELINE configs[4] on one GPU) — the rocprofv3 --kernel-trace --stats summary, the FETCH_SIZE / WRITE_SIZE passes
# (separate, as the guide prescribes) and one SQ pass.  Outputs under gpurun_out/ with the round prefix given as $1 (default r05).
# The profiled commands keep every round's Merkle tree on the main stream (IOPX_MERKLE_STREAM=0, exported here — rocprofv3's own command line stays the
# program itself): with the trees on the side stream two kernels run at once, and a per-kernel duration or counter would be the pair's.
set -u
P=${1:-r06}
R=$PWD
export TMPDIR=/tmp
export IOPX_MERKLE_STREAM=0
cd /tmp
STEPS=5; WARM=2
BENCH="python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary --no-cross-check"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_prof -- $BENCH > $R/gpurun_out/${P}_bench_under_rocprof.json 2> $R/gpurun_out/${P}_prof.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${P}_pmc_fetch -- $BENCH > /dev/null 2> $R/gpurun_out/${P}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${P}_pmc_write -- $BENCH > /dev/null 2> $R/gpurun_out/${P}_pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/${P}_pmc_sq -- $BENCH > /dev/null 2> $R/gpurun_out/${P}_pmc_sq.err
# the native Fractal prover: 3 indexer runs + 4 proofs per command, the last one under the library's own profiler (algorithmic bytes per kernel)
FR="python3 $R/tools/fractal_bench.py --log-n 20 --reps 3 --native --native-only"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_fr_prof -- $FR --out $R/gpurun_out/${P}_fractal_2p20.json > /dev/null 2> $R/gpurun_out/${P}_fr_prof.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${P}_fr_fetch -- $FR > /dev/null 2> $R/gpurun_out/${P}_fr_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${P}_fr_write -- $FR > /dev/null 2> $R/gpurun_out/${P}_fr_write.err
cd $R
# bench.py also proves the warm-up proof of iopx_aurora_instance_warm, a first timed proof, one with the trees on the main stream and one under its own
# profiler: STEPS + WARM + 4 proofs per command
python3 tools/rocprof_summary.py $(find gpurun_out/${P}_prof -name "*.db" | head -1) > gpurun_out/${P}_rocprofv3_bench_aurora2p20.txt
python3 tools/rocprof_summary.py $(find gpurun_out/${P}_fr_prof -name "*.db" | head -1) > gpurun_out/${P}_rocprofv3_fractal2p20.txt
# GPU idle gaps inside the last proof of the traced run (under the tracer's own per-launch overhead: an upper bound on the unprofiled gaps)
# (one proof period of the timed loop: from the 4th proof's lincheck kernel to the 5th's)
python3 tools/gpu_gaps.py gpurun_out/${P}_prof --min-us 30 --period-kernel k_lincheck_add --period-index 5 --histogram > gpurun_out/${P}_gpu_gaps.txt
python3 tools/make_traffic_json.py gpurun_out/${P}_pmc_fetch gpurun_out/${P}_pmc_write 20 gpurun_out/${P}_traffic_aurora.json --bench-json gpurun_out/${P}_bench_under_rocprof.json --steps $((STEPS + WARM + 4)) > /dev/null
python3 tools/make_traffic_json.py gpurun_out/${P}_fr_fetch gpurun_out/${P}_fr_write 20 gpurun_out/${P}_traffic_fractal.json --bench-json gpurun_out/${P}_fractal_2p20.json --steps 4 --min-ms 0.2 > /dev/null
python3 tools/make_sq_json.py gpurun_out/${P}_pmc_sq gpurun_out/${P}_sq_aurora.json k_bfly_upperILb0 k_bfly_edge_multiILb0 k_bfly_edge_fwd_batch k_bfly_upperILb1 k_bfly_edge_multiILb1 k_ldt_combine_add_slots k_merkle_leaves_sub24ILi4 k_merkle_leaves_sub24ILi1ELi2 k_merkle_level k_lincheck_add > /dev/null
rm -rf gpurun_out/${P}_prof gpurun_out/${P}_pmc_fetch gpurun_out/${P}_pmc_write gpurun_out/${P}_pmc_sq gpurun_out/${P}_fr_prof gpurun_out/${P}_fr_fetch gpurun_out/${P}_fr_write
head -14 gpurun_out/${P}_rocprofv3_bench_aurora2p20.txt
