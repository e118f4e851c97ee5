#!/bin/bash
# Same-box A/B of two (or more) builds of the library:
#   gpurun --timeout 900 -- 'CONFIGS="f32 int8" bash tools/gpu/r3_ab.sh TAG rten_amd/_ab/old.so rten_amd/_ab/new.so'
# (box-to-box variance is +-1-2 %, more than most kernel changes: the builds are timed on ONE box, interleaved, twice)
TAG=$1; shift
CONFIGS=${CONFIGS:-int8}
mkdir -p gpurun_out
for cfg in $CONFIGS; do
  for rep in 1 2; do
    for so in "$@"; do
      n=$(basename $so .so)
      RTEN_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 > gpurun_out/${TAG}_${n}_${cfg}_bench$rep.json
      python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_${n}_${cfg}_bench$rep.json"))
print("$cfg $n rep$rep ms_per_step", d["ms_per_step"], "value", d["value"], "logits", d["ranks"]["logits_sha16_per_rank"])
PY
    done
  done
done
if [ -n "$PER_LAYER" ]; then
  for so in "$@"; do
    n=$(basename $so .so)
    RTEN_HIP_LIBRARY=$PWD/$so timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_${n}_per_layer.txt 2>&1
    echo "== $n"; tail -n 2 gpurun_out/${TAG}_${n}_per_layer.txt
  done
fi
