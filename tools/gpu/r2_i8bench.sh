#!/bin/bash
tag=${1:-r2y}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -k "second_stream" 2>&1 | tail -3
for extra in "" "--concurrent"; do
timeout 600 python bench.py --config int8 --steps 50 --warmup 20 --no-cpu-baseline $extra > gpurun_out/${tag}_bench_int8.json 2> gpurun_out/${tag}_bench_int8.err
echo "rc=$? $extra"
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench_int8.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["shortcut_branch"])
PY
done
