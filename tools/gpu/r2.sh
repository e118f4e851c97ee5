#!/bin/bash
# Round-2 GPU session: smoke, GPU parity tests, bench (f32 headline + secondary incl. the int8 line), quantisation probe.
# Usage: gpurun --timeout 2400 -- 'bash tools/gpu/r2.sh [tag] [skip-tests]'
TAG=${1:-r2a}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > gpurun_out/${TAG}_hw.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
if [ -z "$2" ]; then
timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/${TAG}_pytest_r2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r2.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_round2.py > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
fi
timeout 900 python bench.py --layer-table --save-plan gpurun_out/${TAG}_plan.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
timeout 300 python tools/probe_quantization.py 3 > gpurun_out/${TAG}_quant_probe.txt 2>&1
tail -3 gpurun_out/${TAG}_smoke.log; tail -25 gpurun_out/${TAG}_pytest_r2.log; tail -8 gpurun_out/${TAG}_pytest_gpu.log; head -c 1500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_quant_probe.txt
