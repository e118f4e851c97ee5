#!/bin/bash
# Round 6: the expand -> reduce convolution pairs of stage 0 in one launch (rten_hip_conv2d_f32_pair; plan key "pairs"): parity test, the kernel against the two
# launches it replaces (alone and under four-stream co-run), whole-model A/Bs under the lanes schedule and on one replica.
TAG=${1:-r11}
O=gpurun_out/$TAG
mkdir -p $O
P=profiles/plans
timeout 300 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "two_pointwise" 2>&1 | tail -2
timeout 200 python tools/probe_conv_pair.py > $O/conv_pair_probe.txt 2>&1; cat $O/conv_pair_probe.txt
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], d['p50_latency_ms'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes4 committed          " ""
run "f32 lanes4 three pairs        " "--load-plan $P/experiments/f32_lanes_pairs.json"
run "f32 lanes4 two pairs (M2 = 64)" "--load-plan $P/experiments/f32_lanes_pairs2.json"
done 2>&1 | tee $O/f32_pairs_ab.txt
