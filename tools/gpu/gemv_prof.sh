#!/bin/bash
# Kernel-trace timing of the MatMulNBits decode kernel for the (U, SPL) tuning variants (host launch rate hides a < 10 us kernel from the
# event-bracketed microbench).  Usage: tools/gpu/gemv_prof.sh
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/gemv
cd /tmp
for u in 8 32; do for s in 0 2 4 8; do
  RTEN_HIP_GEMV_U=$u RTEN_HIP_GEMV_SPL=$s timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gemv/u${u}_s${s} -o t -- python $R/tools/bench_ops.py --only "NBits decode" > /dev/null 2>&1
  echo "U=$u SPL=$s $(grep gemv4 $R/gpurun_out/gemv/u${u}_s${s}/t_kernel_stats.csv | cut -d, -f1-5 | cut -c1-30,60-)"
done; done
find $R/gpurun_out/gemv -name "t_kernel_trace.csv" -delete
