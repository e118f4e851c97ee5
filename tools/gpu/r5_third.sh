#!/bin/bash
# Round-5 third session: the FULL-shape instantiation of the fused attention (parity + ablation ladder), and a SAME-BOX A/B of the int8 lines between the
# library of session r5a (rten_amd/_ab/librten_hip_r5a.so, built from commit 19c6246) and the current one, interleaved twice.
#   gpurun --timeout 900 -- 'bash tools/gpu/r5_third.sh r5c'
TAG=${1:-r5c}
R=$(pwd)
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest "tests/test_gpu_parity.py::test_sdpa_bit_exact" "tests/test_gpu_parity.py::test_sdpa_head64_shapes_bit_exact" "tests/test_gpu_parity.py::test_bert_encoder_bit_exact" \
   "tests/test_graph_executor.py::test_bert_encoder_onnx_graph_bit_exact" -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 300 python tools/probe_sdpa.py > $O/sdpa_ablation.txt 2>&1
timeout 200 python tools/bench_bert.py --no-cpu-baseline > $O/bench_bert.json 2> $O/bench_bert.err
for rep in 1 2; do
  for lib in r5a cur; do
    so=$R/rten_amd/librten_hip.so; [ $lib = r5a ] && so=$R/rten_amd/_ab/librten_hip_r5a.so
    RTEN_HIP_LIBRARY=$so timeout 200 python bench.py --config int8 --via-runner --no-secondary --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/int8_runner_${lib}_$rep.json
  done
done
timeout 200 python bench.py --config int8 --no-secondary --no-cpu-baseline > $O/bench_int8.json 2> $O/bench_int8.err
tail -n 3 $O/tests.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/int8_runner_*.json"))+["$O/bench_int8.json","$O/bench_bert.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        pool={k:round(v["ms"]/v["launches"]*1e3,1) for k,v in (r.get("other_kernels") or {}).items() if "pool" in k}
        qo={k:round(v["ms"]/v["launches"]*1e3,1) for k,v in (r.get("igemm_i8_family",{}).get("variants") or {}).items()}
        print(f.split("/")[-1], d["ms_per_step"], r["frac"], pool, qo, r.get("fused_attention"))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/sdpa_ablation.txt
