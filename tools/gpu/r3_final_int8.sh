#!/bin/bash
# The int8 half of r3_final.sh (after an int8-only kernel change): smoke, the driver's bench line (its `secondary` carries the int8 run), the int8 bench line,
# per-layer table, rocprofv3 kernel stats, matrix-pipe counters and FETCH / WRITE traffic of the int8 plan.   gpurun --timeout 900 -- 'bash tools/gpu/r3_final_int8.sh r06c'
TAG=${1:-r06c}
R=$(pwd)
P=$R/profiles/plans
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config int8 --no-secondary > gpurun_out/${TAG}_bench_int8.json 2> gpurun_out/${TAG}_bench_int8.err
timeout 200 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_int8_per_layer.txt 2>&1
cd /tmp
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_int8 -o t -- python $R/bench.py --config int8 $COMMON > $R/gpurun_out/${TAG}_prof_int8.json 2> $R/gpurun_out/${TAG}_prof_int8.err
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PMCARGS="--no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_int8 -o t -- python $R/bench.py --config int8 $PMCARGS > $R/gpurun_out/${TAG}_pmc_int8.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_int8_$c -o t -- python $R/bench.py --config int8 $PMCARGS > $R/gpurun_out/${TAG}_int8_$c.log 2>&1
done
cd $R
f() { find gpurun_out/$1 -name "$2" | head -1; }
python tools/pmc_mfma.py $(f ${TAG}_pmc_int8 t_counter_collection.csv) 3 > gpurun_out/${TAG}_mfma_util_int8.csv
python tools/pmc_traffic.py $(f ${TAG}_int8_FETCH_SIZE t_counter_collection.csv) $(f ${TAG}_int8_WRITE_SIZE t_counter_collection.csv) $P/int8.json > gpurun_out/${TAG}_int8_hbm_traffic_per_kernel.json
cp $(f ${TAG}_prof_int8 t_kernel_stats.csv) gpurun_out/${TAG}_rocprofv3_kernel_stats_int8.csv 2>/dev/null
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "t_counter_collection.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
tail -n 2 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
for n in ("bench","bench_int8"):
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("traffic"), d["config"]["launch_plan"]["sha16"], r.get("traffic_plan_sha16"))
        if "secondary" in d: print("  secondary int8", d["secondary"]["resnet50_int8_b32"]["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
tail -n 1 gpurun_out/${TAG}_int8_per_layer.txt
