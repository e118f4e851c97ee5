#!/bin/bash
# Round 4: matrix-pipe counters of the committed one-chain f32 plan (serialised launches: counter passes cannot show the overlapped state).
TAG=${1:-r07}
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PMCARGS="--no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 400 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_f32_1chain -o t -- python $R/bench.py --chains 1 $PMCARGS > $R/gpurun_out/${TAG}_pmc_f32_1chain.log 2>&1
cd $R
python tools/pmc_mfma.py $(find gpurun_out/${TAG}_pmc_f32_1chain -name t_counter_collection.csv | head -1) 3 > gpurun_out/${TAG}_mfma_util_f32_1chain.csv
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "t_counter_collection.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
head -14 gpurun_out/${TAG}_mfma_util_f32_1chain.csv
