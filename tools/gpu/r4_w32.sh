#!/bin/bash
# Round 4: 32x32 barrier-free wave tiles (variants 28, 29): parity on the layer shapes, then the tuner with them among the candidates.
TAG=${1:-w32}
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "all_variants or split_k" 2>&1 | tail -3
bash tools/gpu/r4_tune.sh $TAG
grep "^\[layer\]" gpurun_out/${TAG}_bench_tune_1chain.err | head -60
