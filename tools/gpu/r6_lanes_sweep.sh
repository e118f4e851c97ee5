#!/bin/bash
# Round 6 (second part): the co-run plans as the committed lanes plans -- product-path parity at BASELINE size, then the number of replicas under them.
TAG=${1:-r10e}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model_baseline.py tests/test_gpu_multirank.py -m gpu -x -q > $O/tests_baseline.log 2>&1; tail -4 $O/tests_baseline.log
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
runb() { timeout 300 python tools/bench_bert.py --no-cpu-baseline $2 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/err.txt; }
for rep in 1 2; do
run "f32 lanes2" "--lanes 2"
run "f32 lanes3" "--lanes 3"
run "f32 lanes4" "--lanes 4"
run "f32 lanes5" "--lanes 5"
done 2>&1 | tee $O/f32_lanes.txt
for rep in 1 2; do
runb "bert lanes2" "--lanes 2"
runb "bert lanes3" "--lanes 3"
runb "bert lanes4" "--lanes 4"
runb "bert lanes5" "--lanes 5"
runb "bert lanes6" "--lanes 6"
done 2>&1 | tee $O/bert_lanes.txt
