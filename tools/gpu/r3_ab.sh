#!/bin/bash
# Same-box A/B of two builds of the library: gpurun --timeout 900 -- 'bash tools/gpu/r3_ab.sh TAG rten_amd/_ab/old.so rten_amd/_ab/new.so'
# (box-to-box variance is +-1-2 %, more than most kernel changes: both builds are timed on ONE box, interleaved)
TAG=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  for so in "$@"; do
    n=$(basename $so .so)
    RTEN_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --config int8 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/${TAG}_${n}_bench$rep.json
    python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_${n}_bench$rep.json"))
print("$n rep$rep ms_per_step", d["ms_per_step"], "value", d["value"])
PY
  done
done
for so in "$@"; do
  n=$(basename $so .so)
  RTEN_HIP_LIBRARY=$PWD/$so timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_${n}_per_layer.txt 2>&1
  echo "== $n"; tail -n 2 gpurun_out/${TAG}_${n}_per_layer.txt
done
