#!/bin/bash
tag=${1:-r2y}
mkdir -p gpurun_out
timeout 600 python bench.py --config int8 --steps 50 --warmup 20 --no-cpu-baseline --layer-table > gpurun_out/${tag}_bench_int8.json 2> gpurun_out/${tag}_bench_int8.err
echo "rc=$?"
grep "^\[layer\]" gpurun_out/${tag}_bench_int8.err | cut -c1-150
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench_int8.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step"]["frac"], d["config"]["quantize_on_load_layers"])
PY
timeout 600 python bench.py --config int8 --steps 50 --warmup 20 --no-cpu-baseline --no-autotune > gpurun_out/${tag}_bench_int8_notune.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench_int8_notune.json").read().strip().splitlines()[-1])
print("no autotune (all staged):", d["value"], d["ms_per_step"])
PY
