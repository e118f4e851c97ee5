#!/bin/bash
# Round 6: per-layer int8 tiles chosen under co-run (4 streams for the lanes plan, 1 stream for the one-replica plan), whole-model A/Bs, the knob's test.
TAG=${1:-r10v}
O=gpurun_out/$TAG
mkdir -p $O
P=profiles/plans
timeout 300 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "tile_knob" 2>&1 | tail -2
timeout 600 python tools/tune_corun_int8.py --lanes 4 --plan $P/int8_lanes.json --out $O/int8_lanes_tiles.json > $O/tune_int8_corun4.txt 2> $O/err1.txt; tail -2 $O/err1.txt; cat $O/tune_int8_corun4.txt | cut -c1-200
timeout 600 python tools/tune_corun_int8.py --lanes 1 --plan $P/int8.json --out $O/int8_1lane_tiles.json > $O/tune_int8_1lane.txt 2> $O/err2.txt; tail -2 $O/err2.txt; tail -1 $O/tune_int8_1lane.txt
C="--config int8 --no-secondary --no-cpu-baseline"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['p50_latency_ms'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "int8 lanes4 committed     " ""
run "int8 lanes4 per-layer tile" "--load-plan $O/int8_lanes_tiles.json"
done 2>&1 | tee $O/int8_ab.txt
for rep in 1 2; do
run "int8 lanes1 committed     " "--lanes 1"
run "int8 lanes1 per-layer tile" "--lanes 1 --load-plan $O/int8_1lane_tiles.json"
run "int8 lanes1 lanes-tiles   " "--lanes 1 --load-plan $O/int8_lanes_tiles.json"
done 2>&1 | tee -a $O/int8_ab.txt
