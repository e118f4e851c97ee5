#!/bin/bash
# Round 6: the recompute form of quantized outputs (plan key "qout2"): op-level parity, then the int8 step under lanes with candidate edge sets.
TAG=${1:-r09f}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round6.py -x -q > $O/tests.log 2>&1; tail -12 $O/tests.log
C="--config int8 --no-secondary --no-cpu-baseline"
P=profiles/plans/experiments
run() { # label, extra args
  timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['p50_latency_ms'], f['ranks']['logits_sha16_per_rank'], f['config']['launch_plan']['steps_planned'])" || tail -5 $O/err.txt
}
for rep in 1 2; do
run "lanes4 base          " ""
run "lanes4 qout2 c2      " "--load-plan $P/int8_lanes_qout2_c2.json"
run "lanes4 qout2 c2 s0-1 " "--load-plan $P/int8_lanes_qout2_c2_s01.json"
run "lanes4 qout2 c1+c2   " "--load-plan $P/int8_lanes_qout2_c1c2.json"
run "lanes4 qout2 c1+c2 nd" "--load-plan $P/int8_lanes_qout2_c1c2_nodql.json"
done 2>&1 | tee $O/qout2_lanes.txt
run "lanes1 base (qout)   " "--lanes 1" | tee -a $O/qout2_lanes.txt
run "lanes1 qout2 c1+c2   " "--lanes 1 --load-plan $P/int8_lanes_qout2_c1c2.json" | tee -a $O/qout2_lanes.txt
run "lanes2 base          " "--lanes 2" | tee -a $O/qout2_lanes.txt
run "lanes2 qout2 c1+c2   " "--lanes 2 --load-plan $P/int8_lanes_qout2_c1c2.json" | tee -a $O/qout2_lanes.txt
run "lanes6 qout2 c1+c2   " "--lanes 6 --load-plan $P/int8_lanes_qout2_c1c2.json" | tee -a $O/qout2_lanes.txt
for r in 0 54 80 160; do RTEN_LN_LDS=$r timeout 100 python tools/probe_layer_norm.py >> $O/layer_norm_lds.txt 2>&1; done; cat $O/layer_norm_lds.txt
