#!/bin/bash
# Batch-1 latency (BASELINE configs[0]) of two builds on one box, the headline line beside it, smoke on the default build.
O=$PWD/$1; N=$PWD/$2
mkdir -p gpurun_out
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
for rep in 1 2; do
  for so in $O $N; do
    RTEN_HIP_LIBRARY=$so timeout 40 python tools/bench_resnet50_b1.py --steps 300 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b1 f32', '$(basename $so)', d['p50_latency_ms'], d['value'])"
  done
done
for so in $O $N; do
  RTEN_HIP_LIBRARY=$so timeout 40 python bench.py --steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b32 f32', '$(basename $so)', d['ms_per_step'], d['ranks']['logits_sha16_per_rank'])"
done
timeout 45 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resnet50_batch1 or gemm_f32_bit_exact_size_matrix" 2>&1 | tail -n 2
