#!/bin/bash
# Round-5 experiment: occupancy caps (rten_hip_set_gemm_order bits 4-6: workgroups per compute unit) under the lanes schedule -- do two replicas' launches
# mix better (one's prologue / epilogue beside the other's k-loops) when neither can fill a compute unit alone?
TAG=${1:-r5j}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
C="--no-secondary --no-cpu-baseline --no-shapes"
timeout 200 python bench.py $C > $O/f32_default.json 2> $O/f32_default.err
for cap in 2 3 4; do
  for l in 2 3; do
    timeout 200 python bench.py --lanes $l --load-plan profiles/plans/experiments/f32_1chain_cap$cap.json $C > $O/f32_cap${cap}_lanes$l.json 2> $O/f32_cap${cap}_lanes$l.err
  done
done
timeout 200 python bench.py --lanes 4 --load-plan profiles/plans/experiments/f32_1chain_cap2.json $C > $O/f32_cap2_lanes4.json 2> $O/f32_cap2_lanes4.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("ms_per_step_joined_every_step"), r["frac"], d["ranks"]["logits_sha16_per_rank"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
