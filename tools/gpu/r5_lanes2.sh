#!/bin/bash
# Round-5: the new default schedules (f32: one chain x 2 lanes; int8: 4 lanes) under the driver's flags and the defaults, f32 at 3 lanes, BERT-base at
# 1..3 lanes and as 2 sub-batch chains, the two-rank bench test on the lanes path.
#   gpurun --timeout 1200 -- 'bash tools/gpu/r5_lanes2.sh r5f'
TAG=${1:-r5f}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
C="--no-secondary --no-cpu-baseline --no-shapes"
timeout 200 python bench.py --steps 20 --warmup 5 $C > $O/f32_default_driver_flags.json 2> $O/f32_default_driver_flags.err
timeout 200 python bench.py $C > $O/f32_default.json 2> $O/f32_default.err
timeout 200 python bench.py --lanes 3 $C > $O/f32_lanes3.json 2> $O/f32_lanes3.err
timeout 200 python bench.py --chains 4 $C > $O/f32_4chains_1lane.json 2> $O/f32_4chains_1lane.err
timeout 200 python bench.py --config int8 --steps 20 --warmup 5 $C > $O/int8_default_driver_flags.json 2> $O/int8_default_driver_flags.err
timeout 200 python bench.py --config int8 $C > $O/int8_default.json 2> $O/int8_default.err
for l in 1 2 3; do
  timeout 200 python tools/bench_bert.py --lanes $l --no-cpu-baseline > $O/bert_lanes$l.json 2> $O/bert_lanes$l.err
done
timeout 200 python tools/bench_bert.py --chains 2 --no-cpu-baseline > $O/bert_chains2.json 2> $O/bert_chains2.err
timeout 200 python tools/bench_bert.py --chains 2 --lanes 2 --no-cpu-baseline > $O/bert_chains2_lanes2.json 2> $O/bert_chains2_lanes2.err
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q > $O/tests_multirank.log 2>&1; echo "rc=$?" >> $O/tests_multirank.log; tail -n 3 $O/tests_multirank.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("ms_per_step_joined_every_step"), d.get("p50_latency_ms"), (r.get("step") or r)["frac"], (d["config"].get("batch_lanes") or d["config"].get("launch_plan")))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
