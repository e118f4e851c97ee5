#!/bin/bash
# Round-5 second session: new-kernel parity (streaming max-pool, LDS mask row in the fused attention), the int8 executor after the zero-point fix,
# the attention ablation ladder, pooling microbench A/B.
#   gpurun --timeout 900 -- 'bash tools/gpu/r5_second.sh r5b'
TAG=${1:-r5b}
R=$(pwd)
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py "tests/test_gpu_parity.py::test_sdpa_bit_exact" "tests/test_gpu_parity.py::test_sdpa_head64_shapes_bit_exact" \
   "tests/test_gpu_round3.py::test_max_pool_stats_same_values_and_the_statistics_the_quantizer_would_sweep" \
   "tests/test_graph_executor.py::test_model_abi_int8_quantized_output_edges_from_the_plan_file" "tests/test_graph_executor.py::test_resnet50_int8_onnx_graph_bit_exact" \
   "tests/test_graph_executor.py::test_bert_encoder_onnx_graph_bit_exact" -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 200 python bench.py --config int8 --no-secondary --no-cpu-baseline > $O/bench_int8.json 2> $O/bench_int8.err
timeout 200 python bench.py --config int8 --via-runner --no-secondary --no-cpu-baseline > $O/bench_int8_runner.json 2> $O/bench_int8_runner.err
RTEN_HIP_DEBUG=1048576 timeout 200 python bench.py --config int8 --no-secondary --no-cpu-baseline > $O/bench_int8_oldpool.json 2> $O/bench_int8_oldpool.err
timeout 300 python tools/probe_sdpa.py > $O/sdpa_ablation.txt 2>&1
timeout 200 python tools/bench_bert.py --no-cpu-baseline > $O/bench_bert.json 2> $O/bench_bert.err
timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-shapes > $O/bench_f32.json 2> $O/bench_f32.err
tail -n 3 $O/tests.log
python - <<PY
import json
for n in ("bench_int8","bench_int8_runner","bench_int8_oldpool","bench_bert","bench_f32"):
    try:
        d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["frac"], d["config"].get("path"), (d.get("ranks") or {}).get("logits_sha16_per_rank"), {k:v for k,v in (r.get("other_kernels") or {}).items() if "pool" in k}, r.get("fused_attention"))
    except Exception as e: print(n, "ERR", e)
PY
cat $O/sdpa_ablation.txt
