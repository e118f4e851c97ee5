#!/bin/bash
# Round 6 (second part): per-layer plans chosen under self-co-run (three streams running the same layer), then the whole model A/B against the committed plan.
TAG=${1:-r10c}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python tools/tune_corun.py --lanes 3 --out $O/f32_corun3.json ${TUNE_ARGS:-} > $O/tune_corun3.txt 2> $O/tune_err.txt; tail -3 $O/tune_err.txt; tail -30 $O/tune_corun3.txt | cut -c1-330
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes3 committed   " ""
run "f32 lanes3 corun3 plan " "--load-plan $O/f32_corun3.json"
done 2>&1 | tee $O/f32_ab.txt
run "f32 lanes1 committed   " "--lanes 1" | tee -a $O/f32_ab.txt
run "f32 lanes1 corun3 plan " "--lanes 1 --load-plan $O/f32_corun3.json" | tee -a $O/f32_ab.txt
