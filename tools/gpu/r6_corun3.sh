#!/bin/bash
# Round 6 (second part): the f32 lanes plan re-chosen under FOUR streams (the new default), and the one-replica plans re-chosen by the same tools at one stream
# (full candidate lists incl. persistent / split forms for the MatMul family, which the executor's own tuner does not try); whole-model A/Bs on the same box.
TAG=${1:-r10f}
O=gpurun_out/$TAG
mkdir -p $O
P=profiles/plans
timeout 1500 python tools/tune_corun.py --lanes 4 --full --out $O/f32_corun4_full.json > $O/tune_corun4_full.txt 2> $O/tune_err.txt; tail -3 $O/tune_err.txt; tail -1 $O/tune_corun4_full.txt
timeout 1500 python tools/tune_corun.py --lanes 1 --full --plan $P/f32_1chain.json --out $O/f32_1lane_full.json > $O/tune_1lane_full.txt 2> $O/tune_err1.txt; tail -3 $O/tune_err1.txt; tail -1 $O/tune_1lane_full.txt
timeout 900 python tools/tune_corun_gemm.py --lanes 1 --plan $P/bert_base_b32_s128.json --out $O/bert_1lane.json > $O/tune_bert_1lane.txt 2> $O/tune_bert_err.txt; tail -3 $O/tune_bert_err.txt; cat $O/tune_bert_1lane.txt | cut -c1-300
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
runb() { timeout 300 python tools/bench_bert.py --no-cpu-baseline $2 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes4 committed (3-stream plan)" ""
run "f32 lanes4 4-stream plan            " "--load-plan $O/f32_corun4_full.json"
done 2>&1 | tee $O/f32_ab.txt
for rep in 1 2; do
run "f32 lanes1 committed                " "--lanes 1"
run "f32 lanes1 re-chosen plan           " "--lanes 1 --load-plan $O/f32_1lane_full.json"
runb "bert lanes1 committed              " "--lanes 1"
runb "bert lanes1 re-chosen plan         " "--lanes 1 --load-plan $O/bert_1lane.json"
done 2>&1 | tee $O/one_lane_ab.txt
