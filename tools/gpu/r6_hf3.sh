#!/bin/bash
TAG=${1:-r10h}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_shape_arithmetic.py tests/test_gpu_round6.py -m gpu -x -q > $O/tests_shape.log 2>&1; tail -5 $O/tests_shape.log
timeout 600 python tools/bench_bert.py --hf --lanes 1 > $O/bench_bert_hf_export_1lane.json 2> $O/bench_bert_hf_export_1lane.err; tail -c 300 $O/bench_bert_hf_export_1lane.err
timeout 600 python tools/bench_bert.py --hf --lanes 4 > $O/bench_bert_hf_export.json 2> $O/bench_bert_hf_export.err; tail -c 300 $O/bench_bert_hf_export.err
python - <<PY
import json
for n in ("bench_bert_hf_export_1lane","bench_bert_hf_export"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["roofline"]["frac"], d["kernels"], d["config"]["launch_plan"].get("hf_export"))
PY
timeout 1700 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; tail -5 $O/gputests.log
