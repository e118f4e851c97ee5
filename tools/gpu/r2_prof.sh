#!/bin/bash
# Round-2 profiling session on the FINAL kernels: sustained-MFMA probe, rocprofv3 kernel stats of the bench command (f32 and
# int8), matrix-pipe PMC pass (kernel-trace only: no sys / hip traces with --pmc), HBM traffic passes.
# Usage: gpurun --timeout 2400 -- 'bash tools/gpu/r2_prof.sh <tag>'
TAG=${1:-r05}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_sustained $R/tools/probes/mfma_sustained.hip && timeout 300 /tmp/mfma_sustained > $R/gpurun_out/${TAG}_mfma_sustained.txt 2>&1
cd $R
# headline bench line + plan (autotune once; every profiled run below replays this plan)
timeout 900 python bench.py --layer-table --save-plan gpurun_out/${TAG}_plan.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32 -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --load-plan $R/gpurun_out/${TAG}_plan.json > $R/gpurun_out/${TAG}_prof_f32.json 2> $R/gpurun_out/${TAG}_prof_f32.err
# the same forward pass as ONE chain (kernel durations that do not overlap each other: the per-kernel cross-check for bench.py's roofline object)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32_1chain -o t -- python $R/bench.py --chains 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof_f32_1chain.json 2> $R/gpurun_out/${TAG}_prof_f32_1chain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_int8 -o t -- python $R/bench.py --config int8 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof_int8.json 2> $R/gpurun_out/${TAG}_prof_int8.err
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_f32 -o t -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --load-plan $R/gpurun_out/${TAG}_plan.json > $R/gpurun_out/${TAG}_pmc_f32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_int8 -o t -- python $R/bench.py --config int8 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_int8.log 2>&1
cd $R
python tools/pmc_mfma.py $(find gpurun_out/${TAG}_pmc_f32 -name "t_counter_collection.csv" | head -1) 3 > gpurun_out/${TAG}_mfma_util_f32.csv
python tools/pmc_mfma.py $(find gpurun_out/${TAG}_pmc_int8 -name "t_counter_collection.csv" | head -1) 3 > gpurun_out/${TAG}_mfma_util_int8.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_f32_$c -o t -- python $R/bench.py --load-plan $R/gpurun_out/${TAG}_plan.json --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_f32_$c.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_int8_$c -o t -- python $R/bench.py --config int8 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_int8_$c.log 2>&1)
done
python tools/pmc_traffic.py $(find gpurun_out/${TAG}_f32_FETCH_SIZE -name "t_counter_collection.csv" | head -1) $(find gpurun_out/${TAG}_f32_WRITE_SIZE -name "t_counter_collection.csv" | head -1) > gpurun_out/${TAG}_hbm_traffic_per_kernel.json
python tools/pmc_traffic.py $(find gpurun_out/${TAG}_int8_FETCH_SIZE -name "t_counter_collection.csv" | head -1) $(find gpurun_out/${TAG}_int8_WRITE_SIZE -name "t_counter_collection.csv" | head -1) > gpurun_out/${TAG}_int8_hbm_traffic_per_kernel.json
for d in prof_f32 prof_f32_1chain prof_int8; do cp $(find gpurun_out/${TAG}_$d -name "t_kernel_stats.csv" | head -1) gpurun_out/${TAG}_${d}_kernel_stats.csv; done
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "t_counter_collection.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
du -sh gpurun_out
cat gpurun_out/${TAG}_mfma_sustained.txt; head -12 gpurun_out/${TAG}_mfma_util_f32.csv; head -12 gpurun_out/${TAG}_mfma_util_int8.csv; head -14 gpurun_out/${TAG}_prof_int8_kernel_stats.csv | cut -c1-160
