#!/bin/bash
# Round-3 profile collection on the GPU box (run from the repo root through gpurun): the rocprofv3 --kernel-trace --stats summary of
# the bench command, the FETCH_SIZE / WRITE_SIZE passes (separate, as the guide prescribes) and one SQ pass.  Outputs under gpurun_out/.
set -u
R=$PWD
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof -- $BENCH > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_prof.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r03_pmc_fetch -- $BENCH > /dev/null 2> $R/gpurun_out/r03_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r03_pmc_write -- $BENCH > /dev/null 2> $R/gpurun_out/r03_pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/r03_pmc_sq -- $BENCH > /dev/null 2> $R/gpurun_out/r03_pmc_sq.err
cd $R
python3 tools/rocprof_summary.py $(find gpurun_out/r03_prof -name "*.db" | head -1) > gpurun_out/r03_rocprofv3_bench_aurora2p20.txt
python3 tools/make_traffic_json.py gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write 20 gpurun_out/r03_traffic_aurora.json k_bfly_upper k_bfly_edge k_phase1 k_ldt_combine_add_slots k_merkle_leaves k_lincheck_add k_fri_fold_fused > /dev/null
python3 tools/make_sq_json.py gpurun_out/r03_pmc_sq gpurun_out/r03_sq_aurora.json k_bfly_upper k_bfly_edge k_phase1 k_ldt_combine_add_slots k_merkle_leaves k_lincheck_add > /dev/null
rm -rf gpurun_out/r03_prof gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write gpurun_out/r03_pmc_sq
head -12 gpurun_out/r03_rocprofv3_bench_aurora2p20.txt
