#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench (with per-layer table), rocprof kernel stats.
# Usage: gpurun --timeout 2400 -- 'bash tools/gpu/round.sh [tag]'
TAG=${1:-r01}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > gpurun_out/${TAG}_hw.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --layer-table --save-plan gpurun_out/${TAG}_plan.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --load-plan $R/gpurun_out/${TAG}_plan.json > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err; echo "rocprof rc=$?" >> $R/gpurun_out/${TAG}_prof.err
cd $R
(rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|Counter_Name|gpu|  *[A-Z_0-9]+)" | head -400) > gpurun_out/${TAG}_counters.txt 2>&1
find gpurun_out/${TAG}_prof -name "*stats*" | head; du -sh gpurun_out
# keep only the small stats summaries (the full kernel trace can be large)
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -size +2M -delete
tail -3 gpurun_out/${TAG}_smoke.log; tail -15 gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_bench.json | head -c 3000; tail -5 gpurun_out/${TAG}_bench.err
