#!/bin/bash
# per-layer A/B timing of launch plans: tools/gpu/r2_layers.sh <tag> <layers> <plans>
TAG=${1:-r2e}
mkdir -p gpurun_out
timeout 600 python tools/layer_probe.py --layers "$2" --variants "$3" --reps 20 > gpurun_out/${TAG}_layers.txt 2>&1
cat gpurun_out/${TAG}_layers.txt
