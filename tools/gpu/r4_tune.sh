#!/bin/bash
# Round 4: re-tune the f32 launch plans with the wave-tile variants (24..26) among the candidates.   gpurun --timeout 1800 -- 'bash tools/gpu/r4_tune.sh w1'
TAG=${1:-w1}
mkdir -p gpurun_out/${TAG}_plans
timeout 700 python bench.py --autotune --chains 1 --layer-table --save-plan gpurun_out/${TAG}_plans/f32_1chain.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_tune_1chain.json 2> gpurun_out/${TAG}_bench_tune_1chain.err
timeout 900 python bench.py --autotune --layer-table --save-plan gpurun_out/${TAG}_plans/f32_4chains.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_tune.json 2> gpurun_out/${TAG}_bench_tune.err
python - <<PY
import json
for n in ["bench_tune_1chain","bench_tune"]:
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(n, "ERR", e)
PY
grep -c "^\[layer\]" gpurun_out/${TAG}_bench_tune_1chain.err
