O=gpurun_out/r11; mkdir -p $O; P=profiles/plans
timeout 300 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "stem or two_pointwise" 2>&1 | tail -2
timeout 200 python tools/probe_conv_pair.py 2>&1 | grep streams
timeout 300 python tools/probe_stem.py 2>&1 | grep "32, 0\|27, 0"
C="--no-secondary --no-cpu-baseline --no-shapes"
run() { timeout 200 python bench.py $C $2 --detail-file $O/d.json 2>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/d.json')); print('$1', d['ms_per_step'], d['roofline']['frac'], f['ranks']['logits_sha16_per_rank'])" || tail -3 $O/err.txt; }
for rep in 1 2 3; do
run "f32 lanes4 committed (pairs)  " ""
run "f32 lanes4 pairs + stem direct" "--load-plan $P/experiments/f32_lanes_stem32.json"
done 2>&1 | tee $O/f32_stem_ab2.txt
