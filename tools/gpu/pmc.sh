#!/bin/bash
# PMC passes (separate runs per counter set; kernel-trace only, no sys/hip traces) over isolated layers.
# Usage: tools/gpu/pmc.sh <tag> <layers> <variants|-> [plan.json]
TAG=${1:-pmc}
LAYERS=${2:-s0b0c2,s0b0c3,s2b1c1,s2b1c2,s3b1c2}
VARS=${3:-0,3}
PLAN=${4:-}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--layers $LAYERS --reps 3"
if [ -n "$PLAN" ]; then ARGS="$ARGS --plan $R/$PLAN"; else ARGS="$ARGS --variants $VARS"; fi
cd /tmp
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}_$n -o $n -- python $R/tools/layer_probe.py $ARGS > $R/gpurun_out/${TAG}_$n.log 2>&1
  echo "$n rc=$?" >> $R/gpurun_out/${TAG}_$n.log
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
ls gpurun_out/${TAG}_*/ | head -20
find gpurun_out -name "*kernel_trace*" -size +3M -delete
tail -12 gpurun_out/${TAG}_sq1.log
