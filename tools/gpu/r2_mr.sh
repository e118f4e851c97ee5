#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q --tb=short -p no:cacheprovider 2>&1 | tail -15
