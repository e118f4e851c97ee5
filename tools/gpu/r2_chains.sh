#!/bin/bash
# chained runner: parity test + bench (default = 4 chains) vs --chains 1, 2
tag=${1:-r2x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -k "chained" 2>&1 | tail -5
for c in 4 2 1; do
  timeout 600 python bench.py --steps 50 --warmup 20 --chains $c --no-cpu-baseline --no-secondary > gpurun_out/${tag}_bench_c$c.json 2> gpurun_out/${tag}_bench_c$c.err
  echo "bench c=$c rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench_c$c.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], d["config"]["batch_chains"], r["kernel"], r["frac"], r["igemm_family"]["frac"], r.get("step"))
PY
done
