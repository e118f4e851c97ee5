#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "one_row or batch1_logits or matmul_ops or gemm_f32 or einsum" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_cpp_host.py tests/test_einsum.py -x -q --tb=short -m gpu 2>&1 | tail -5
timeout 300 python tools/bench_resnet50_b1.py 2>/dev/null | tail -1 | cut -c1-200
