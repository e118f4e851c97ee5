#!/bin/bash
# ablation of the persistent kernel's k-loop (librten_hip_ablate.so, -DRTEN_ABLATE): RTEN_HIP_DEBUG bits 1 = no in-loop DMA,
# 2 = no MFMA / fragment reads, 8 = no barrier, 16 = MFMAs on register operands, 32 = no vmcnt wait
TAG=${1:-r2i}
mkdir -p gpurun_out
: > gpurun_out/${TAG}_ablate.txt
for dbg in $4; do
  echo "== RTEN_HIP_DEBUG=$dbg" >> gpurun_out/${TAG}_ablate.txt
  RTEN_HIP_DEBUG=$dbg RTEN_HIP_LIBRARY=$(pwd)/rten_amd/librten_hip_ablate.so timeout 300 python tools/layer_probe.py --layers "$2" --variants "$3" --reps 10 >> gpurun_out/${TAG}_ablate.txt 2>&1
done
cat gpurun_out/${TAG}_ablate.txt
