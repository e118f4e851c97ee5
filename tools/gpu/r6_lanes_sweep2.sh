O=gpurun_out/r11; mkdir -p $O
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/err.txt; }
for rep in 1 2; do
for l in 3 4 5 6; do run "f32 lanes $l" "--lanes $l"; done
done 2>&1 | tee $O/f32_lanes_sweep_final_plan.txt
