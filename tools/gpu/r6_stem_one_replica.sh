#!/bin/bash
# Round 6: the direct stem kernel on ONE replica (1 chain, 4 chains of 8 images): whole-model A/B.
O=gpurun_out/r11; mkdir -p $O; P=profiles/plans
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], d['p50_latency_ms'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 1 chain 1 lane committed  " "--lanes 1"
run "f32 1 chain 1 lane stem direct" "--lanes 1 --load-plan $P/experiments/f32_1chain_stem32.json"
run "f32 4 chains committed        " "--chains 4 --lanes 1"
run "f32 4 chains stem direct      " "--chains 4 --lanes 1 --load-plan $P/experiments/f32_4chains_stem32.json"
done 2>&1 | tee $O/f32_stem_one_replica_ab.txt
