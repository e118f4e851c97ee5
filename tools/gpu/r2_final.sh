#!/bin/bash
# Round-2 closing session on the final code: smoke, all GPU tests, the three bench lines (f32 4 chains = the driver's command, f32 one chain,
# int8), matrix-pipe counters of the one-chain plan, the per-layer int8 table.   gpurun --timeout 2400 -- 'bash tools/gpu/r2_final.sh r05f'
TAG=${1:-r05f}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > gpurun_out/${TAG}_hw.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/${TAG}_pytest_r2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_r2.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_round2.py > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --layer-table --save-plan gpurun_out/${TAG}_plan.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --chains 1 --no-secondary --save-plan gpurun_out/${TAG}_plan_1chain.json > gpurun_out/${TAG}_bench_1chain.json 2> gpurun_out/${TAG}_bench_1chain.err
timeout 600 python bench.py --config int8 --no-secondary > gpurun_out/${TAG}_bench_int8.json 2> gpurun_out/${TAG}_bench_int8.err
timeout 300 python tools/probe_int8_per_layer.py > gpurun_out/${TAG}_int8_per_layer.txt 2>&1
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/${TAG}_pmc_f32_1chain -o t -- python $R/bench.py --chains 1 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --load-plan $R/gpurun_out/${TAG}_plan_1chain.json > $R/gpurun_out/${TAG}_pmc_f32_1chain.log 2>&1
cd $R
python tools/pmc_mfma.py $(find gpurun_out/${TAG}_pmc_f32_1chain -name "t_counter_collection.csv" | head -1) 3 > gpurun_out/${TAG}_mfma_util_f32_1chain.csv
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "t_counter_collection.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
tail -2 gpurun_out/${TAG}_smoke.log; tail -3 gpurun_out/${TAG}_pytest_r2.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log
python - <<PY
import json
for n in ("bench","bench_1chain","bench_int8"):
    try:
        d=json.loads(open("gpurun_out/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("igemm_family",{}).get("frac"), r.get("step",{}).get("frac"), r.get("traffic"), r.get("traffic_source"))
    except Exception as e: print(n, "ERR", e)
PY
head -12 gpurun_out/${TAG}_mfma_util_f32_1chain.csv; tail -2 gpurun_out/${TAG}_int8_per_layer.txt
