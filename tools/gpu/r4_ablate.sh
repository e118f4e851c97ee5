#!/bin/bash
# Round 4: layer-level ablation ladder of the production f32 kernel (an -DRTEN_ABLATE build, rten_amd/_ab/ablate.so): what does a real layer
# cost without its in-loop DMA / MFMAs / epilogue / barrier?   gpurun --timeout 600 -- 'bash tools/gpu/r4_ablate.sh a1'
TAG=${1:-a1}
mkdir -p gpurun_out
L=s0b1c1,s0b1c2,s0b1c3,s1b0c1,s1b1c2,s1b1c3,s2b1c1,s2b1c2,s3b1c2
for dbg in 0 1 2 4 8 16 5 20 21; do
  echo "== RTEN_HIP_DEBUG=$dbg (1 no in-loop DMA, 2 no MFMA, 4 no epilogue, 8 no barrier, 16 MFMA on register operands)"
  RTEN_HIP_DEBUG=$dbg RTEN_HIP_LIBRARY=$PWD/rten_amd/_ab/ablate.so timeout 120 python tools/layer_probe.py --layers $L --variants 3:0:1 --reps 10 2>&1 | grep plan
done | tee gpurun_out/${TAG}_ablate_ladder.txt
