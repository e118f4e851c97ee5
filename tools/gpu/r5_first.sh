#!/bin/bash
# Round-5 first session: the product path at BASELINE size against the oracle (new tests), the default bench lines (executor), the BERT plan tuned through
# the executor, the relaxed-vs-strict split-K probe, kernel stats of the int8 executor and runner for the gap.  Every command under its own timeout.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r5_first.sh r5a'
TAG=${1:-r5a}
R=$(pwd)
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > $O/hw.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_model_baseline.py "tests/test_graph_executor.py::test_model_abi_chains_and_plan_file_give_the_oracle_bits" \
   "tests/test_graph_executor.py::test_model_abi_int8_quantized_output_edges_from_the_plan_file" -x -q > $O/tests_model.log 2>&1; echo "rc=$?" >> $O/tests_model.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-shapes > $O/bench_50.json 2> $O/bench_50.err
timeout 200 python bench.py --via-runner --no-secondary --no-cpu-baseline > $O/bench_runner.json 2> $O/bench_runner.err
timeout 200 python bench.py --config int8 --no-secondary --no-cpu-baseline > $O/bench_int8.json 2> $O/bench_int8.err
timeout 200 python bench.py --config int8 --via-runner --no-secondary --no-cpu-baseline > $O/bench_int8_runner.json 2> $O/bench_int8_runner.err
timeout 300 python tools/bench_bert.py --autotune --save-plan $O/bert_base_b32_s128.json --no-cpu-baseline > $O/bench_bert_tuned.json 2> $O/bench_bert_tuned.err
timeout 200 python tools/bench_bert.py --via-runner --no-cpu-baseline > $O/bench_bert_runner.json 2> $O/bench_bert_runner.err
timeout 200 python tools/probe_relaxed_split.py --batch 32 > $O/relaxed_split_b32.txt 2>&1
timeout 200 python tools/probe_relaxed_split.py --batch 8 > $O/relaxed_split_b8.txt 2>&1
cd /tmp
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-shapes"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_int8 -o t -- python $R/bench.py --config int8 $COMMON > $R/$O/prof_int8.json 2> $R/$O/prof_int8.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_int8_runner -o t -- python $R/bench.py --config int8 --via-runner $COMMON > $R/$O/prof_int8_runner.json 2> $R/$O/prof_int8_runner.err
cd $R
f() { find $O/$1 -name "$2" | head -1; }
for n in int8 int8_runner; do cp $(f prof_$n t_kernel_stats.csv) $O/rocprofv3_kernel_stats_$n.csv 2>/dev/null; done
find $O -name "t_kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q > $O/tests_multirank.log 2>&1; echo "rc=$?" >> $O/tests_multirank.log
tail -n 3 $O/smoke.log $O/tests_model.log $O/tests_multirank.log
python - <<PY
import json
for n in ("bench","bench_50","bench_runner","bench_int8","bench_int8_runner","bench_bert_tuned","bench_bert_runner"):
    try:
        d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], d.get("ms_per_step_joined_every_step"), r.get("kernel"), r["frac"], d["config"].get("path"), d["ranks"]["logits_sha16_per_rank"] if "ranks" in d else "")
    except Exception as e: print(n, "ERR", e)
PY
cat $O/relaxed_split_b32.txt | tail -12
