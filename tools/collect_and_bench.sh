#!/bin/bash
# One GPU-box call for a round's final numbers: the rocprofv3 passes (tools/collect_profiles.sh), their summaries copied into profiles/ so that
# bench.py quotes the PMC traffic of the sources it runs, then the benches.  Outputs under gpurun_out/ (copy what is to be judged into profiles/).
set -u
P=${1:-r06}
bash tools/collect_profiles.sh $P > gpurun_out/${P}_collect.log 2>&1
for f in traffic_aurora.json traffic_fractal.json sq_aurora.json rocprofv3_bench_aurora2p20.txt rocprofv3_fractal2p20.txt gpu_gaps.txt bench_under_rocprof.json fractal_2p20.json; do
    cp gpurun_out/${P}_$f profiles/${P}_$f
done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${P}_f_bench_aurora_1gpu.json 2> gpurun_out/${P}_f_bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-sharded --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${P}_f_bench_aurora_forcesharded_1rank_rccl.json 2> gpurun_out/${P}_f_fs.err
python3 tools/fractal_bench.py --native --reps 3 --out gpurun_out/${P}_f_fractal_native_1gpu.json > gpurun_out/${P}_f_fractal.log 2>&1
python3 tools/native_bench.py --field edwards_Fr --out gpurun_out/${P}_f_native_aurora_edwards.json > /dev/null 2>&1
python3 - <<PY
import json
d = json.loads(open("gpurun_out/${P}_f_bench_aurora_1gpu.json").read().strip().splitlines()[-1])
print("aurora ms/step", d["ms_per_step"], "traffic", d["roofline"]["traffic"], "ref schedule", d["config"]["reference_schedule"]["ms_per_step"], "fractal", d["config"]["secondary_fractal"]["prover_ms_min"])
PY
