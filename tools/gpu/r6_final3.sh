#!/bin/bash
# Round-6 closing session, third part (pair kernel, stem kernel in the lanes / four-chain plans): smoke, the DRIVER command with raw stdout kept, the other schedules, int8 / BERT lines, ops microbench,
# rocprofv3 kernel stats, FETCH / WRITE traffic passes stamped with the plan hashes, matrix-pipe counters, the driver command again with the traffic attached.
# Every command under its own timeout.        gpurun --timeout 2400 -- 'bash tools/gpu/r6_final2.sh r10'
TAG=${1:-r11f}
R=$(pwd)
P=$R/profiles/plans
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > $O/hardware.txt 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.stdout 2> $O/driver_cmd.stderr; echo "rc=$?" >> $O/driver_cmd.stderr
for f in bench_detail.json bench_detail_int8.json bench_detail_f32_4chains.json; do cp gpurun_out/$f $O/$f 2>/dev/null; done
C="--no-secondary --no-cpu-baseline --no-shapes"
timeout 200 python bench.py $C --detail-file $O/bench_default_50_20_detail.json > $O/bench_default_50_20.json 2> $O/bench_default_50_20.err
timeout 200 python bench.py --lanes 1 $C --detail-file $O/bench_1chain_1lane_detail.json > $O/bench_1chain_1lane.json 2> $O/bench_1chain_1lane.err
timeout 200 python bench.py --chains 4 $C --detail-file $O/bench_4chains_1lane_detail.json > $O/bench_4chains_1lane.json 2> $O/bench_4chains_1lane.err
timeout 200 python bench.py --config int8 --lanes 1 $C --detail-file $O/bench_int8_1lane_detail.json > $O/bench_int8_1lane.json 2> $O/bench_int8_1lane.err
timeout 300 python tools/bench_bert.py > $O/bench_bert.json 2> $O/bench_bert.err
timeout 200 python tools/bench_bert.py --lanes 1 --no-cpu-baseline > $O/bench_bert_1lane.json 2> $O/bench_bert_1lane.err
timeout 600 python tools/bench_bert.py --hf --lanes 1 > $O/bench_bert_hf_export_1lane.json 2> $O/bench_bert_hf_export_1lane.err
timeout 600 python tools/bench_bert.py --hf --lanes 4 > $O/bench_bert_hf_export.json 2> $O/bench_bert_hf_export.err
timeout 400 python tools/bench_ops.py > $O/ops_microbench.json 2> $O/ops_microbench.err
cd /tmp
PC="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-shapes"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32 -o t -- python $R/bench.py $PC > $R/$O/prof_f32.json 2> $R/$O/prof_f32.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_int8 -o t -- python $R/bench.py --config int8 $PC > $R/$O/prof_int8.json 2> $R/$O/prof_int8.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bert -o t -- python $R/tools/bench_bert.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bert.json 2> $R/$O/prof_bert.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32_1lane -o t -- python $R/bench.py --lanes 1 $PC > $R/$O/prof_f32_1lane.json 2> $R/$O/prof_f32_1lane.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_int8_1lane -o t -- python $R/bench.py --config int8 --lanes 1 $PC > $R/$O/prof_int8_1lane.json 2> $R/$O/prof_int8_1lane.err
PMCARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-shapes"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/f32_$c -o t -- python $R/bench.py --lanes 1 --load-plan $P/f32_lanes.json $PMCARGS > $R/$O/f32_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/int8_$c -o t -- python $R/bench.py --config int8 --lanes 2 $PMCARGS > $R/$O/int8_$c.log 2>&1
done
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/$O/pmc_f32 -o t -- python $R/bench.py --lanes 1 --load-plan $P/f32_lanes.json $PMCARGS > $R/$O/pmc_f32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/$O/pmc_int8 -o t -- python $R/bench.py --config int8 --lanes 2 $PMCARGS > $R/$O/pmc_int8.log 2>&1
cd $R
f() { find $O/$1 -name "$2" | head -1; }
python tools/pmc_traffic.py $(f f32_FETCH_SIZE t_counter_collection.csv) $(f f32_WRITE_SIZE t_counter_collection.csv) $P/f32_lanes.json > $O/hbm_traffic_per_kernel.json
python tools/pmc_traffic.py $(f int8_FETCH_SIZE t_counter_collection.csv) $(f int8_WRITE_SIZE t_counter_collection.csv) $P/int8_lanes.json > $O/int8_hbm_traffic_per_kernel.json
python tools/pmc_mfma.py $(f pmc_f32 t_counter_collection.csv) 3 > $O/mfma_util_f32.csv
python tools/pmc_mfma.py $(f pmc_int8 t_counter_collection.csv) 3 > $O/mfma_util_int8.csv
for n in f32 int8 bert f32_1lane int8_1lane; do cp $(f prof_$n t_kernel_stats.csv) $O/rocprofv3_kernel_stats_$n.csv 2>/dev/null; done
find $O -name "t_kernel_trace.csv" -size +2M -delete; find $O -name "t_counter_collection.csv" -size +4M -delete; find $O -name "*.db" -delete
# the driver's command once more, now that the PMC traffic of THIS plan hash exists (bench.py attaches roofline.traffic only from a pass over the plan it times)
mkdir -p profiles/r11; cp $O/hbm_traffic_per_kernel.json $O/int8_hbm_traffic_per_kernel.json profiles/r11/
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_final.stdout 2> $O/driver_cmd_final.stderr; echo "rc=$?" >> $O/driver_cmd_final.stderr
cp gpurun_out/bench_detail.json $O/bench_detail_final.json 2>/dev/null
for i in 1 2 3; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-shapes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fresh process', $i, d['ms_per_step'], d['roofline']['frac'])"; done > $O/repeatability_driver_flags.txt 2>&1
tail -n 2 $O/smoke.log
wc -c $O/driver_cmd.stdout $O/driver_cmd_final.stdout; tail -c 4200 $O/driver_cmd_final.stdout; cat $O/repeatability_driver_flags.txt
python - <<PY
import json
for n in ("bench_default_50_20","bench_1chain_1lane","bench_4chains_1lane","bench_int8_1lane","bench_bert","bench_bert_1lane","bench_bert_hf_export_1lane","bench_bert_hf_export"):
    try:
        d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], d.get("p50_latency_ms"), r.get("kernel"), r.get("frac"), r.get("traffic"), (d.get("config") or {}).get("launch_plan"))
    except Exception as e:
        print(n, "failed:", e)
PY
