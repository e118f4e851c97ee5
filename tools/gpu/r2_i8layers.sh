#!/bin/bash
tag=${1:-r2y}
mkdir -p gpurun_out
timeout 600 python tools/probe_int8_per_layer.py > gpurun_out/${tag}_int8_per_layer.txt 2>&1
echo rc=$?
cat gpurun_out/${tag}_int8_per_layer.txt
