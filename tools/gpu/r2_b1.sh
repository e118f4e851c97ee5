#!/bin/bash
tag=${1:-r2b1}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -k "last_arrival" 2>&1 | tail -2
timeout 300 python tools/bench_resnet50_b1.py > gpurun_out/${tag}_b1.json 2>/dev/null
RTEN_HIP_DEBUG=1048576 timeout 300 python tools/bench_resnet50_b1.py > gpurun_out/${tag}_b1_smallfold.json 2>/dev/null
timeout 300 python tools/bench_resnet50_b1.py --config int8 > gpurun_out/${tag}_b1_int8.json 2>/dev/null
timeout 300 python tools/bench_resnet50_b1.py --batch 8 > gpurun_out/${tag}_b8.json 2>/dev/null
python - <<PY
import json
for n in ("b1","b1_smallfold","b1_int8","b8"):
    d=json.loads(open("gpurun_out/${tag}_%s.json"%n).read().strip().splitlines()[-1]); print(n, {k:d[k] for k in ("value","ms_per_step_back_to_back","p50_latency_ms") if k in d})
PY
