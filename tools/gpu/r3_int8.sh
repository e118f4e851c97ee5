#!/bin/bash
# Round-3 int8 session: the quantized-output tests, the per-layer table (with the one-launch conv + consumer's quantize column), the int8
# bench line with the feature off and on.   gpurun --timeout 1200 -- 'bash tools/gpu/r3_int8.sh r3a'
TAG=${1:-r3a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -s --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_r3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r3.log
timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_int8_per_layer.txt 2>&1
timeout 400 python bench.py --config int8 --no-secondary --no-cpu-baseline --no-qout > gpurun_out/${TAG}_bench_int8_noqout.json 2> gpurun_out/${TAG}_bench_int8_noqout.err
timeout 400 python bench.py --config int8 --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_int8.json 2> gpurun_out/${TAG}_bench_int8.err
tail -n 12 gpurun_out/${TAG}_pytest_r3.log; tail -n 3 gpurun_out/${TAG}_int8_per_layer.txt
python - <<PY
import json
for n in ("bench_int8_noqout","bench_int8"):
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
