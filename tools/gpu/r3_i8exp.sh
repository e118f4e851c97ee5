#!/bin/bash
# int8 kernel experiments: the per-layer table under RTEN_HIP_DEBUG tuning switches.  gpurun --timeout 1200 -- 'bash tools/gpu/r3_i8exp.sh r3e 0 4096 8192 16384'
TAG=$1; shift
mkdir -p gpurun_out
for dbg in "$@"; do
  RTEN_HIP_DEBUG=$dbg timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_int8_per_layer_dbg$dbg.txt 2>&1
  echo "== RTEN_HIP_DEBUG=$dbg"; tail -n 1 gpurun_out/${TAG}_int8_per_layer_dbg$dbg.txt
done
