#!/bin/bash
# Round 4: epilogue-load prefetch (order bit 3 = off) per layer and per step.   gpurun --timeout 900 -- 'bash tools/gpu/r4_pre.sh p1'
TAG=${1:-p1}
mkdir -p gpurun_out
L=s0b0c1,s0b1c1,s0b1c2,s0b1c3,s1b0c1,s1b1c1,s1b1c2,s1b1c3,s2b1c1,s2b1c2,s2b1c3,s3b1c1,s3b1c3
python tools/layer_probe.py --layers $L --variants 3:0:1:0,3:0:1:8,15:0:1:0,15:0:1:8 --reps 10 2>&1 | grep plan | tee gpurun_out/${TAG}_layers.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_f32_resnet_layer_shapes or conv_f32_residual or conv_f32_split" 2>&1 | tail -3
for ch in 4 1; do
  timeout 300 python bench.py --chains $ch --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch ON chains $ch', d['ms_per_step'], d['ranks']['logits_sha16_per_rank'])"
done
