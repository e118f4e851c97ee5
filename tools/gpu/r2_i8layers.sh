#!/bin/bash
tag=${1:-r2y}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "int8 or integer or Integer or quant" 2>&1 | tail -3
for dbg in 0 2048; do
  echo "=== RTEN_HIP_DEBUG=$dbg"
  RTEN_HIP_DEBUG=$dbg timeout 300 python tools/probe_int8_per_layer.py 2>&1
done > gpurun_out/${tag}_int8_kg.txt 2>&1
grep -c . gpurun_out/${tag}_int8_kg.txt
