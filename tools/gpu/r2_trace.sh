#!/bin/bash
# cycle trace of the persistent kernel's k-loop (instrumented build librten_hip_trace.so): tools/gpu/r2_trace.sh <tag> <layers> <plans>
TAG=${1:-r2f}
mkdir -p gpurun_out
RTEN_HIP_LIBRARY=$(pwd)/rten_amd/librten_hip_trace.so timeout 600 python tools/layer_probe.py --layers "$2" --variants "$3" --reps 2 > gpurun_out/${TAG}_trace.txt 2>&1
cat gpurun_out/${TAG}_trace.txt
