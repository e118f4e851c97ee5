#!/bin/bash
# rocprofv3 kernel stats of the one-chain f32 plan (replayed from profiles/r05/plan_1chain.json: no autotune launches in the trace)
TAG=${1:-r05g}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32_1chain -o t -- python $R/bench.py --chains 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --load-plan $R/profiles/r05/plan_1chain.json > $R/gpurun_out/${TAG}_prof_f32_1chain.json 2> $R/gpurun_out/${TAG}_prof_f32_1chain.err
cd $R
cp $(find gpurun_out/${TAG}_prof_f32_1chain -name "t_kernel_stats.csv" | head -1) gpurun_out/${TAG}_prof_f32_1chain_kernel_stats.csv
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*.db" -delete
head -8 gpurun_out/${TAG}_prof_f32_1chain_kernel_stats.csv | cut -c1-200
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_prof_f32_1chain.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["launches"], r["frac"])
PY
