#!/bin/bash
# Round 6: the first block's shortcut convolution inside its pair (rten_hip_conv2d_f32_pair_shortcut; plan key "pair_shortcuts"): tests, the kernel against what it
# replaces, whole-model A/Bs.
O=gpurun_out/r11; mkdir -p $O; P=profiles/plans
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_graph_executor.py -m gpu -x -q -k "shortcut or two_pointwise or chains_and_plan" 2>&1 | tail -3
timeout 300 python tools/probe_conv_pair_shortcut.py > $O/conv_pair_shortcut_probe.txt 2>&1; tail -3 $O/conv_pair_shortcut_probe.txt
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], d['p50_latency_ms'], f['ranks']['logits_sha16_per_rank'], f['config']['launch_plan'].get('steps_planned'))" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes4 committed       " ""
run "f32 lanes4 + shortcut      " "--load-plan $P/experiments/f32_lanes_shortcut.json"
run "f32 4 chains committed     " "--chains 4 --lanes 1"
run "f32 4 chains + shortcut    " "--chains 4 --lanes 1 --load-plan $P/experiments/f32_4chains_shortcut.json"
done 2>&1 | tee $O/f32_pair_shortcut_ab.txt
