O=gpurun_out/r11; mkdir -p $O; P=profiles/plans
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'], f['config']['launch_plan'].get('steps_planned'))" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes4 shortcut + three pairs (third: 256 -> 128)" "--load-plan $P/experiments/f32_lanes_shortcut.json"
run "f32 lanes4 shortcut + two pairs                      " "--load-plan $P/experiments/f32_lanes_shortcut_2pairs.json"
done 2>&1 | tee $O/f32_third_pair_ab.txt
