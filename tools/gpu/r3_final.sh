#!/bin/bash
# Round-3 closing session on the final code with the committed launch plans (profiles/plans/): smoke, the three bench lines (f32 4 chains =
# the driver's command, f32 one chain, int8), rocprofv3 kernel stats of the same commands, matrix-pipe counters of the one-chain plan,
# FETCH / WRITE traffic passes of the 4-chain and int8 plans (stamped with the plan hash), per-layer int8 table, ops microbench.
#   gpurun --timeout 3000 -- 'bash tools/gpu/r3_final.sh r06'        (the test suite runs separately: tools/gpu/r3_test.sh)
TAG=${1:-r06}
R=$(pwd)
P=$R/profiles/plans
mkdir -p gpurun_out
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > gpurun_out/${TAG}_hw.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --chains 1 --no-secondary > gpurun_out/${TAG}_bench_1chain.json 2> gpurun_out/${TAG}_bench_1chain.err
timeout 600 python bench.py --config int8 --no-secondary > gpurun_out/${TAG}_bench_int8.json 2> gpurun_out/${TAG}_bench_int8.err
timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_int8_per_layer.txt 2>&1
timeout 600 python tools/bench_ops.py > gpurun_out/${TAG}_ops_microbench.json 2> gpurun_out/${TAG}_ops_microbench.err
cd /tmp
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32 -o t -- python $R/bench.py $COMMON > $R/gpurun_out/${TAG}_prof_f32.json 2> $R/gpurun_out/${TAG}_prof_f32.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_f32_1chain -o t -- python $R/bench.py --chains 1 $COMMON > $R/gpurun_out/${TAG}_prof_f32_1chain.json 2> $R/gpurun_out/${TAG}_prof_f32_1chain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_int8 -o t -- python $R/bench.py --config int8 $COMMON > $R/gpurun_out/${TAG}_prof_int8.json 2> $R/gpurun_out/${TAG}_prof_int8.err
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PMCARGS="--no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_f32_1chain -o t -- python $R/bench.py --chains 1 $PMCARGS > $R/gpurun_out/${TAG}_pmc_f32_1chain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_int8 -o t -- python $R/bench.py --config int8 $PMCARGS > $R/gpurun_out/${TAG}_pmc_int8.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_f32_$c -o t -- python $R/bench.py $PMCARGS > $R/gpurun_out/${TAG}_f32_$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_int8_$c -o t -- python $R/bench.py --config int8 $PMCARGS > $R/gpurun_out/${TAG}_int8_$c.log 2>&1
done
cd $R
f() { find gpurun_out/$1 -name "$2" | head -1; }
python tools/pmc_mfma.py $(f ${TAG}_pmc_f32_1chain t_counter_collection.csv) 3 > gpurun_out/${TAG}_mfma_util_f32_1chain.csv
python tools/pmc_mfma.py $(f ${TAG}_pmc_int8 t_counter_collection.csv) 3 > gpurun_out/${TAG}_mfma_util_int8.csv
python tools/pmc_traffic.py $(f ${TAG}_f32_FETCH_SIZE t_counter_collection.csv) $(f ${TAG}_f32_WRITE_SIZE t_counter_collection.csv) $P/f32_4chains.json > gpurun_out/${TAG}_hbm_traffic_per_kernel.json
python tools/pmc_traffic.py $(f ${TAG}_int8_FETCH_SIZE t_counter_collection.csv) $(f ${TAG}_int8_WRITE_SIZE t_counter_collection.csv) $P/int8.json > gpurun_out/${TAG}_int8_hbm_traffic_per_kernel.json
for n in f32 f32_1chain int8; do cp $(f ${TAG}_prof_$n t_kernel_stats.csv) gpurun_out/${TAG}_rocprofv3_kernel_stats_$n.csv 2>/dev/null; done
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "t_counter_collection.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
tail -n 2 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
for n in ("bench","bench_1chain","bench_int8"):
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("traffic"), d["config"]["launch_plan"])
    except Exception as e: print(n, "ERR", e)
PY
head -8 gpurun_out/${TAG}_mfma_util_f32_1chain.csv; tail -n 2 gpurun_out/${TAG}_int8_per_layer.txt; ls gpurun_out | grep ${TAG}_ | head -40
