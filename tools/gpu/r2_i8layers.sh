#!/bin/bash
# per-layer table of the int8 ResNet-50 pipeline (staging kernel, conv kernel, and the quantize-on-load form where it applies):
#   gpurun --timeout 600 -- 'bash tools/gpu/r2_i8layers.sh <tag>'       RTEN_HIP_DEBUG=2048 in the environment turns the k-groups off
tag=${1:-r2y}
mkdir -p gpurun_out
timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${tag}_int8_per_layer.txt 2>&1
echo rc=$?
cat gpurun_out/${tag}_int8_per_layer.txt
