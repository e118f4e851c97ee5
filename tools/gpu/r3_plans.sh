#!/bin/bash
# Round 3: tune the launch plans ONCE (f32 4 chains = the driver's command, f32 one chain, int8) and write them where bench.py picks them
# up by default (profiles/plans/ -- copy them there from gpurun_out/ and commit), then check that the default command reproduces across
# fresh processes without any stream-placement search.   gpurun --timeout 2400 -- 'bash tools/gpu/r3_plans.sh r3p'
TAG=${1:-r3p}
mkdir -p gpurun_out/${TAG}_plans
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_r3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r3.log
timeout 900 python bench.py --autotune --layer-table --save-plan gpurun_out/${TAG}_plans/f32_4chains.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_tune.json 2> gpurun_out/${TAG}_bench_tune.err
timeout 600 python bench.py --autotune --chains 1 --save-plan gpurun_out/${TAG}_plans/f32_1chain.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_tune_1chain.json 2> gpurun_out/${TAG}_bench_tune_1chain.err
timeout 600 python bench.py --autotune --config int8 --layer-table --save-plan gpurun_out/${TAG}_plans/int8.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_tune_int8.json 2> gpurun_out/${TAG}_bench_tune_int8.err
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --load-plan gpurun_out/${TAG}_plans/f32_4chains.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_repeat_$i.json 2> gpurun_out/${TAG}_repeat_$i.err
done
timeout 300 python bench.py --config int8 --load-plan gpurun_out/${TAG}_plans/int8.json --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_int8.json 2> gpurun_out/${TAG}_bench_int8.err
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_multirank.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_multirank.log
tail -n 5 gpurun_out/${TAG}_pytest_r3.log; tail -n 5 gpurun_out/${TAG}_pytest_multirank.log
python - <<PY
import json
for n in ["bench_tune","bench_tune_1chain","bench_tune_int8","bench_int8"]+["repeat_%d"%i for i in range(1,6)]:
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["launch_plan"])
    except Exception as e: print(n, "ERR", e)
PY
