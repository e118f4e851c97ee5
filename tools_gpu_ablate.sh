#!/bin/bash
mkdir -p gpurun_out
L=${1:-s0b0c2,s1b0c1,s1b1c2}
V=${2:-3,0}
for d in 5 13 21 29; do echo "== RTEN_HIP_DEBUG=$d"; RTEN_HIP_DEBUG=$d python tools/layer_probe.py --layers $L --variants $V --reps 5; done 2>&1 | tee gpurun_out/ablate3.log
