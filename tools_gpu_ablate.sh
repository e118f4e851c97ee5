#!/bin/bash
# Ablation timing of the wave-specialised kernel: full / no-DMA / no-MFMA
mkdir -p gpurun_out
L=${1:-s3b1c2,s2b1c2,s2b1c1,s0b0c2,s0b0c3}
for d in 0 1 2; do echo "== RTEN_HIP_DEBUG=$d"; RTEN_HIP_DEBUG=$d python tools/layer_probe.py --layers $L --variants 8,11 --reps 5; done 2>&1 | tee gpurun_out/ablate.log
