#!/bin/bash
TAG=${1:-r10i}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_shape_arithmetic.py tests/test_gpu_model_baseline.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for rep in 1 2; do
timeout 600 python tools/bench_bert.py --hf --lanes 4 > $O/bench_bert_hf_export.json 2> $O/bench_bert_hf_export.err; tail -c 300 $O/bench_bert_hf_export.err
python - <<PY
import json
for n in ("bench_bert_hf_export",):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["roofline"]["frac"], d["kernels"], d["config"]["launch_plan"])
PY
done
timeout 300 python tools/bench_bert.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own writer lanes4', d['ms_per_step'], d['roofline']['frac'])"
