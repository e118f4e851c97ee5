#!/bin/bash
# Round 4: the product path (C++ executor behind the C ABI) against the Python runner, BERT line, int8 regression tests.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r4_exec.sh e1'
TAG=${1:-e1}
mkdir -p gpurun_out
for cfg in "--chains 4" "--chains 1" "--config int8"; do
  n=$(echo $cfg | tr -d ' -')
  timeout 300 python bench.py $cfg --via-executor --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_exec_${n}.json 2> gpurun_out/${TAG}_exec_${n}.err; echo "executor $cfg rc=$?"
  timeout 300 python bench.py $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_runner_${n}.json 2> gpurun_out/${TAG}_runner_${n}.err; echo "runner $cfg rc=$?"
done
timeout 300 python tools/bench_bert.py > gpurun_out/${TAG}_bert.json 2> gpurun_out/${TAG}_bert.err; echo "bert rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d["roofline"].get("frac"), d.get("ranks",{}).get("logits_sha16_per_rank"), d["config"].get("launch_plan"))
    except Exception as e: print(f, "ERR", e, open(f.replace('.json','.err')).read()[-600:])
PY
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_graph_executor.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
