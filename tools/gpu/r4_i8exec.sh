#!/bin/bash
# Round 4: the int8 graph through the C++ executor with the committed edge plan (quantized-output launches), against the runner.
TAG=${1:-i8x}
timeout 300 python -m pytest tests/test_graph_executor.py -m gpu -x -q -k "int8" 2>&1 | tail -5
timeout 200 python bench.py --config int8 --via-executor --no-cpu-baseline > gpurun_out/${TAG}_exec_plan.json 2> gpurun_out/${TAG}_exec_plan.err
timeout 200 python bench.py --config int8 --via-executor --no-autotune --no-cpu-baseline > gpurun_out/${TAG}_exec_noplan.json 2> gpurun_out/${TAG}_exec_noplan.err
timeout 200 python bench.py --config int8 --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_runner.json 2> gpurun_out/${TAG}_runner.err
python - <<PY
import json
for n in ["exec_plan","exec_noplan","runner"]:
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["ranks"]["logits_sha16_per_rank"], d["config"].get("launch_plan"))
    except Exception as e: print(n, "ERR", e); print(open("gpurun_out/${TAG}_%s.err"%n).read()[-800:])
PY
