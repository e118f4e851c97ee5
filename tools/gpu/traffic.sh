#!/bin/bash
# HBM traffic per kernel of the bench workload: two PMC passes (FETCH_SIZE, WRITE_SIZE) over an eager replay of a tuned plan.
# Usage: tools/gpu/traffic.sh <tag> <plan.json (tracked path)>
TAG=${1:-traffic}
PLAN=${2:-profiles/r02/plan.json}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_$c -o t -- python $R/bench.py --load-plan $R/$PLAN --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_traffic.py gpurun_out/${TAG}_FETCH_SIZE/t_counter_collection.csv gpurun_out/${TAG}_WRITE_SIZE/t_counter_collection.csv > gpurun_out/${TAG}_hbm_traffic_per_kernel.json
find gpurun_out -name "t_kernel_trace.csv" -size +3M -delete; find gpurun_out -name "t_counter_collection.csv" -size +8M -delete
head -c 1500 gpurun_out/${TAG}_hbm_traffic_per_kernel.json
