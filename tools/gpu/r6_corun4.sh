#!/bin/bash
# Round 6: the four-stream plan once more without the alias variants, then the whole-model A/B against the committed plan (same box, alternating).
TAG=${1:-r10l}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python tools/tune_corun.py --lanes 4 --full --exclude 28,29,31 --out $O/f32_corun4_noalias.json > $O/tune_corun4_noalias.txt 2> $O/tune_err.txt; tail -3 $O/tune_err.txt; tail -1 $O/tune_corun4_noalias.txt
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes4 committed         " ""
run "f32 lanes4 re-chosen no alias" "--load-plan $O/f32_corun4_noalias.json"
done 2>&1 | tee $O/f32_ab.txt
