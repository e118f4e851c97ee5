#!/bin/bash
# Round 4, session 3: occupancy cap of the LDS-DMA kernels (RTEN_HIP_OCC_CAP = workgroups per compute unit), step time + workgroup trace.
#   gpurun --timeout 900 -- 'bash tools/gpu/r4_occ.sh t3'
TAG=${1:-t3}
mkdir -p gpurun_out
for cap in 0 3 2; do
  for ch in 1 4; do
    RTEN_HIP_OCC_CAP=$cap RTEN_HIP_LIBRARY=$PWD/rten_amd/_ab/trace.so timeout 300 python tools/debug/f32_trace.py --chains $ch --out gpurun_out/${TAG}_cap${cap}_f32_trace > gpurun_out/${TAG}_cap${cap}_trace_${ch}ch.txt 2>&1
    echo "== cap $cap chains $ch"; grep "^\[trace\]" gpurun_out/${TAG}_cap${cap}_trace_${ch}ch.txt | grep -v "prologue\|first k-tile\|epilogue\|store drain\|distinct"
  done
done
