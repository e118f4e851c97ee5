#!/bin/bash
# Shorter GPU session: GPU parity tests + bench with the per-layer table (no rocprof).
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu/quick.sh <tag> [pytest -k expr]'
TAG=${1:-q}
KEXPR=${2:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$KEXPR" > gpurun_out/${TAG}_pytest_gpu.log 2>&1
else
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1
fi
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --layer-table --save-plan gpurun_out/${TAG}_plan.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
tail -15 gpurun_out/${TAG}_pytest_gpu.log; head -c 2500 gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
