#!/bin/bash
# Round 4, session 1: where does the f32 step's time go?  Wall-clock stamps of every workgroup (tools/debug/f32_trace.py, -DRTEN_TRACE build in rten_amd/_ab/trace.so).
#   gpurun --timeout 900 -- 'bash tools/gpu/r4_trace.sh t1'
TAG=${1:-t1}
mkdir -p gpurun_out
#timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 > gpurun_out/${TAG}_bench_f32.json
#python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_f32.json')); print('product f32 4 chains ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
for ch in 4 1; do
  RTEN_HIP_LIBRARY=$PWD/rten_amd/_ab/trace.so timeout 300 python tools/debug/f32_trace.py --chains $ch --out gpurun_out/${TAG}_f32_trace > gpurun_out/${TAG}_trace_${ch}ch.txt 2>&1
  grep "^\[trace\]" gpurun_out/${TAG}_trace_${ch}ch.txt
done
tail -n 5 gpurun_out/${TAG}_trace_1ch.txt
