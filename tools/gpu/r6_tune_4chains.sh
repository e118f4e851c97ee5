#!/bin/bash
# Round 6: the four-chain plan (one replica, 4 sub-batches of 8 images on 4 streams) re-chosen per layer under four-stream self-co-run at batch 8 -- the state its
# launches run in -- and the whole-model A/B.
O=gpurun_out/r11; mkdir -p $O; P=profiles/plans
python - <<PY
import json
p = json.load(open("$P/f32_4chains.json"))
json.dump(p["8"], open("$O/f32_4chains_flat.json", "w"))
PY
timeout 1500 python tools/tune_corun.py --lanes 4 --batch 8 --full --exclude 28,29,31 --plan $O/f32_4chains_flat.json --out $O/f32_4chains_corun_flat.json > $O/tune_4chains_corun4_batch8.txt 2> $O/tune_err.txt; tail -2 $O/tune_err.txt; cut -c1-150 $O/tune_4chains_corun4_batch8.txt
python - <<PY
import json
p = json.load(open("$P/f32_4chains.json"))
p["8"] = json.load(open("$O/f32_4chains_corun_flat.json"))
json.dump(p, open("$O/f32_4chains_corun.json", "w"))
PY
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], d['p50_latency_ms'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 4 chains committed      " "--chains 4 --lanes 1"
run "f32 4 chains co-run re-tuned" "--chains 4 --lanes 1 --load-plan $O/f32_4chains_corun.json"
done 2>&1 | tee $O/f32_4chains_corun_ab.txt
