#!/bin/bash
# One kernel trace of the bench command; GPU idle gaps of one proof period (tools/gpu_gaps.py) and the consumers of the constant-carrying launches.
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tg_prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $R/gpurun_out/tg.err
cd $R
python3 tools/gpu_gaps.py gpurun_out/tg_prof --min-us 60 --context 3 --period-kernel k_lincheck_add --period-index 3 --histogram > gpurun_out/r04_gpu_gaps.txt
python3 tools/upload_consumers.py gpurun_out/tg_prof > gpurun_out/r04_upload_consumers.txt
rm -rf gpurun_out/tg_prof
head -20 gpurun_out/r04_gpu_gaps.txt; cat gpurun_out/r04_upload_consumers.txt
