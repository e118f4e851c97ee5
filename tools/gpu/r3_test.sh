#!/bin/bash
# Round 3: the whole GPU test suite (round-3 file first, fail fast there), C++ host test included.   gpurun --timeout 2400 -- 'bash tools/gpu/r3_test.sh r3t'
TAG=${1:-r3t}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round3.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_r3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r3.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_round3.py > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -n 8 gpurun_out/${TAG}_pytest_r3.log; tail -n 8 gpurun_out/${TAG}_pytest_gpu.log
