#!/bin/bash
# Round-5: lanes as REPLICAS that share one weight set (rten_hip_model_clone); the f32 one-chain plan under 2 lanes against two hand-made variants (no
# split-K anywhere / split-K only where it parks every depth block); BERT at 4 and 6 lanes; parity of a cloned model.
#   gpurun --timeout 1200 -- 'bash tools/gpu/r5_lanes3.sh r5g'
TAG=${1:-r5g}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
C="--no-secondary --no-cpu-baseline --no-shapes"
timeout 600 python -m pytest tests/test_gpu_model_baseline.py -x -q > $O/tests_model.log 2>&1; echo "rc=$?" >> $O/tests_model.log; tail -n 3 $O/tests_model.log
timeout 200 python bench.py $C > $O/f32_default.json 2> $O/f32_default.err
timeout 200 python bench.py --load-plan profiles/plans/experiments/f32_1chain_nosplit.json $C > $O/f32_nosplit.json 2> $O/f32_nosplit.err
timeout 200 python bench.py --load-plan profiles/plans/experiments/f32_1chain_hybrid.json $C > $O/f32_hybrid.json 2> $O/f32_hybrid.err
timeout 200 python bench.py --steps 20 --warmup 5 $C > $O/f32_default_driver_flags.json 2> $O/f32_default_driver_flags.err
timeout 200 python bench.py --config int8 $C > $O/int8_default.json 2> $O/int8_default.err
timeout 200 python bench.py --config int8 --lanes 6 $C > $O/int8_lanes6.json 2> $O/int8_lanes6.err
for l in 4 6; do
  timeout 200 python tools/bench_bert.py --lanes $l --no-cpu-baseline > $O/bert_lanes$l.json 2> $O/bert_lanes$l.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("ms_per_step_joined_every_step"), d.get("p50_latency_ms"), (r.get("step") or r)["frac"], (d["config"].get("batch_lanes") or {}).get("lanes"), d["config"].get("launch_plan",{}).get("source"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
