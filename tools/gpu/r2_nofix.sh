#!/bin/bash
tag=${1:-r2n}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "split or last_arrival" 2>&1 | tail -3
timeout 600 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_a.json 2>/dev/null
RTEN_HIP_DEBUG=524288 timeout 600 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_b.json 2>/dev/null
timeout 300 python tools/bench_resnet50_b1.py > gpurun_out/${tag}_b1.json 2>/dev/null
RTEN_HIP_DEBUG=524288 timeout 300 python tools/bench_resnet50_b1.py > gpurun_out/${tag}_b1_fixup.json 2>/dev/null
python - <<PY
import json
for n in "ab":
    d=json.loads(open("gpurun_out/${tag}_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"])
for n in ("b1","b1_fixup"):
    d=json.loads(open("gpurun_out/${tag}_%s.json"%n).read().strip().splitlines()[-1]); print(n, {k:d[k] for k in ("value","ms_per_step","p50_latency_ms") if k in d})
PY
