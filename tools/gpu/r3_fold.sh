#!/bin/bash
# The two launch folds of the int8 graph (scale products in the quantizer's launch, max-pool statistics), same box, interleaved.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "products or max_pool_stats or resnet" 2>&1 | tail -n 5
for rep in 1 2; do
  for nf in 1 0; do
    RTEN_INT8_NO_FOLD=$nf timeout 300 python bench.py --config int8 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 > gpurun_out/fold_nf${nf}_$rep.json
    python - <<PY
import json
d = json.load(open("gpurun_out/fold_nf${nf}_$rep.json"))
print("no_fold=$nf rep$rep ms_per_step", d["ms_per_step"], "logits", d["ranks"]["logits_sha16_per_rank"])
PY
  done
done
