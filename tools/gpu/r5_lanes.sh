#!/bin/bash
# Round-5: consecutive batches on independent replicas ("lanes") -- int8 cannot be split into sub-batch chains (its quantizers span the batch), but
# step k + 1 can run beside step k.  int8 at 1..4 lanes, f32 (4 chains) at 2 lanes, f32 2 chains x 2 lanes.
#   gpurun --timeout 900 -- 'bash tools/gpu/r5_lanes.sh r5e'
TAG=${1:-r5e}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
C="--no-secondary --no-cpu-baseline --no-shapes"
timeout 600 python -m pytest tests/test_graph_executor.py -m gpu -x -q > $O/tests_graph_executor.log 2>&1; echo "rc=$?" >> $O/tests_graph_executor.log; tail -n 3 $O/tests_graph_executor.log
for l in 1 2 3 4; do
  timeout 200 python bench.py --config int8 --lanes $l $C > $O/int8_lanes$l.json 2> $O/int8_lanes$l.err
done
timeout 200 python bench.py --config int8 --lanes 2 --steps 20 --warmup 5 $C > $O/int8_lanes2_driver_flags.json 2> $O/int8_lanes2_driver_flags.err
timeout 200 python bench.py --lanes 2 $C > $O/f32_4chains_lanes2.json 2> $O/f32_4chains_lanes2.err
timeout 200 python bench.py --chains 2 --lanes 2 --no-autotune $C > $O/f32_2chains_lanes2_noplan.json 2> $O/f32_2chains_lanes2_noplan.err
timeout 200 python bench.py --chains 1 --lanes 4 $C > $O/f32_1chain_lanes4.json 2> $O/f32_1chain_lanes4.err
timeout 200 python bench.py --chains 1 --lanes 2 $C > $O/f32_1chain_lanes2.json 2> $O/f32_1chain_lanes2.err
timeout 200 python bench.py $C > $O/f32_default.json 2> $O/f32_default.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("ms_per_step_joined_every_step"), d.get("p50_latency_ms"), (r.get("step") or r)["frac"], d["config"]["batch_lanes"]["lanes"], d["ranks"]["logits_sha16_per_rank"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
