#!/bin/bash
# Round 6: cross-workgroup K split of the int8 convolution kernel: stand-alone layer times and output hashes per RTEN_I8_KS setting, then the int8 step with the rule on / off.
TAG=${1:-r09d}
O=gpurun_out/$TAG
mkdir -p $O
for s in 0 RULE 2,3 4,3 2,1 4,1 2,0 4,0 3,3 3,1; do
  if [ "$s" = RULE ]; then timeout 200 python tools/probe_int8_ks.py >> $O/ks_probe.txt 2>&1; else RTEN_I8_KS=$s timeout 200 python tools/probe_int8_ks.py >> $O/ks_probe.txt 2>&1; fi
done
cat $O/ks_probe.txt
C="--config int8 --no-secondary --no-cpu-baseline"
for s in 0 RULE 0 RULE; do
  if [ "$s" = RULE ]; then e=""; else e="RTEN_I8_KS=$s"; fi
  env $e timeout 200 python bench.py $C --lanes 1 --detail-file $O/d1.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes1 KS=$s', d['ms_per_step'], d['p50_latency_ms'])"
  env $e timeout 200 python bench.py $C --detail-file $O/d4.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes4 KS=$s', d['ms_per_step'])"
done 2>&1 | tee $O/ks_step.txt
timeout 600 python -m pytest tests/test_shape_arithmetic.py tests/test_cpp_host.py tests/test_gpu_model_baseline.py -m gpu -x -q 2>&1 | tail -15
