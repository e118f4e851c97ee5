#!/bin/bash
# Round 6: the whole GPU suite, then f32 schedule / plan A/Bs under the default lanes schedule (same box, alternating).
TAG=${1:-r09g}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; tail -6 $O/gputests.log
C="--no-secondary --no-cpu-baseline --no-shapes"
P=profiles/plans/experiments
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes2 committed      " ""
run "f32 lanes2 s2c2 128x64 g9 " "--load-plan $P/f32_lanes_s2c2_128x64.json"
run "f32 lanes2 s2s3c2 128x64  " "--load-plan $P/f32_lanes_s2s3c2_128x64.json"
run "f32 lanes3 committed      " "--lanes 3"
done 2>&1 | tee $O/f32_ab.txt
run "f32 lanes4 committed      " "--lanes 4" | tee -a $O/f32_ab.txt
