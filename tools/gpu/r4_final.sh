#!/bin/bash
# Round-4 closing session on the final code with the committed launch plans (profiles/plans/): smoke, the bench lines (f32 4 chains = the driver's
# command, f32 one chain, int8, the three through the executor behind the C ABI, BERT), ops microbench, rocprofv3 kernel stats of the bench commands,
# FETCH / WRITE traffic passes of the 4-chain plan stamped with the plan hash.  Every command under its own timeout.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r4_final.sh r07'        (the test suite runs separately: timeout 600 python -m pytest tests -m gpu -x -q)
TAG=${1:-r07}
R=$(pwd)
P=$R/profiles/plans
mkdir -p gpurun_out
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > gpurun_out/${TAG}_hw.txt 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
timeout 200 python bench.py --chains 1 --no-secondary > gpurun_out/${TAG}_bench_1chain.json 2> gpurun_out/${TAG}_bench_1chain.err
timeout 200 python bench.py --config int8 --no-secondary > gpurun_out/${TAG}_bench_int8.json 2> gpurun_out/${TAG}_bench_int8.err
timeout 200 python bench.py --via-executor --no-cpu-baseline > gpurun_out/${TAG}_bench_executor.json 2> gpurun_out/${TAG}_bench_executor.err
timeout 200 python bench.py --via-executor --chains 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_executor_1chain.json 2> gpurun_out/${TAG}_bench_executor_1chain.err
timeout 200 python bench.py --via-executor --config int8 --no-cpu-baseline > gpurun_out/${TAG}_bench_executor_int8.json 2> gpurun_out/${TAG}_bench_executor_int8.err
timeout 300 python tools/bench_bert.py > gpurun_out/${TAG}_bench_bert.json 2> gpurun_out/${TAG}_bench_bert.err
timeout 400 python tools/bench_ops.py > gpurun_out/${TAG}_ops_microbench.json 2> gpurun_out/${TAG}_ops_microbench.err
cd /tmp
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32 -o t -- python $R/bench.py $COMMON > $R/gpurun_out/${TAG}_prof_f32.json 2> $R/gpurun_out/${TAG}_prof_f32.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32_1chain -o t -- python $R/bench.py --chains 1 $COMMON > $R/gpurun_out/${TAG}_prof_f32_1chain.json 2> $R/gpurun_out/${TAG}_prof_f32_1chain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_int8 -o t -- python $R/bench.py --config int8 $COMMON > $R/gpurun_out/${TAG}_prof_int8.json 2> $R/gpurun_out/${TAG}_prof_int8.err
PMCARGS="--no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_f32_$c -o t -- python $R/bench.py $PMCARGS > $R/gpurun_out/${TAG}_f32_$c.log 2>&1
done
cd $R
f() { find gpurun_out/$1 -name "$2" | head -1; }
python tools/pmc_traffic.py $(f ${TAG}_f32_FETCH_SIZE t_counter_collection.csv) $(f ${TAG}_f32_WRITE_SIZE t_counter_collection.csv) $P/f32_4chains.json > gpurun_out/${TAG}_hbm_traffic_per_kernel.json
for n in f32 f32_1chain int8; do cp $(f ${TAG}_prof_$n t_kernel_stats.csv) gpurun_out/${TAG}_rocprofv3_kernel_stats_$n.csv 2>/dev/null; done
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "t_counter_collection.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
tail -n 2 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
for n in ("bench","bench_1chain","bench_int8","bench_executor","bench_executor_1chain","bench_executor_int8","bench_bert"):
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r.get("kernel"), r["frac"], r.get("traffic"), d["config"].get("launch_plan"))
    except Exception as e: print(n, "ERR", e)
PY
ls gpurun_out | grep ${TAG}_ | head -40
