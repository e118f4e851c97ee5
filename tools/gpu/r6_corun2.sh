#!/bin/bash
# Round 6 (second part): the co-run tuners over their full candidate lists, then whole-model A/Bs against the committed plans (same box, alternating).
TAG=${1:-r10d}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python tools/tune_corun.py --lanes 3 --full --out $O/f32_corun3_full.json > $O/tune_corun3_full.txt 2> $O/tune_err.txt; tail -3 $O/tune_err.txt; tail -26 $O/tune_corun3_full.txt | cut -c1-250
timeout 900 python tools/tune_corun_gemm.py --lanes 4 --out $O/bert_corun4.json > $O/tune_bert_corun4.txt 2> $O/tune_bert_err.txt; tail -3 $O/tune_bert_err.txt; cat $O/tune_bert_corun4.txt | cut -c1-300
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
runb() { timeout 300 python tools/bench_bert.py --no-cpu-baseline $2 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes3 committed        " ""
run "f32 lanes3 corun3 short list" "--load-plan profiles/plans/experiments/f32_corun3.json"
run "f32 lanes3 corun3 full list " "--load-plan $O/f32_corun3_full.json"
done 2>&1 | tee $O/f32_ab.txt
for rep in 1 2; do
runb "bert lanes4 committed  " ""
runb "bert lanes4 corun4 plan" "--load-plan $O/bert_corun4.json"
done 2>&1 | tee $O/bert_ab.txt
runb "bert lanes1 committed  " "--lanes 1" | tee -a $O/bert_ab.txt
runb "bert lanes1 corun4 plan" "--lanes 1 --load-plan $O/bert_corun4.json" | tee -a $O/bert_ab.txt
