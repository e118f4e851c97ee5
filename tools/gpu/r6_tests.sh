#!/bin/bash
# Round 6: new tests first (fail fast on what changed), then the whole GPU suite.
TAG=${1:-r09c}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_shape_arithmetic.py tests/test_gpu_model_baseline.py tests/test_gpu_multirank.py -m gpu -x -q > $O/new_tests.log 2>&1; tail -25 $O/new_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -8 $O/gputests.log
