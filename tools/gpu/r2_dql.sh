#!/bin/bash
tag=${1:-r2dq}
mkdir -p gpurun_out
timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${tag}_int8_per_layer.txt 2>&1
grep "fused\|sum" gpurun_out/${tag}_int8_per_layer.txt | cut -c1-40,95-200
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -k "dql or int8_batch32_bit_exact_runner" --tb=short 2>&1 | tail -5
