#!/bin/bash
tag=${1:-r2p}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -k "chained" 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_$i.json").read().strip().splitlines()[-1]); c=d["config"]["batch_chains"]
print(d["value"], d["ms_per_step"], c["placement"], c["placement_ms"])
PY
done
