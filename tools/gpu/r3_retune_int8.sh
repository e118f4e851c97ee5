#!/bin/bash
# Re-tune the int8 launch plan (per-edge quantized-output choice, per-layer quantize-on-load choice) on the current kernels and compare it
# with the committed plan on the same box.   gpurun --timeout 900 -- 'bash tools/gpu/r3_retune_int8.sh'
mkdir -p gpurun_out/retune
timeout 600 python bench.py --autotune --config int8 --layer-table --save-plan gpurun_out/retune/int8.json --no-secondary --no-cpu-baseline --steps 200 > gpurun_out/retune/tune.json 2> gpurun_out/retune/tune.err
for rep in 1 2; do
  timeout 300 python bench.py --config int8 --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | tail -n 1 > gpurun_out/retune/committed_$rep.json
  timeout 300 python bench.py --config int8 --load-plan gpurun_out/retune/int8.json --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | tail -n 1 > gpurun_out/retune/new_$rep.json
done
python - <<PY
import json
for n in ["tune", "committed_1", "new_1", "committed_2", "new_2"]:
    try:
        d = json.loads(open("gpurun_out/retune/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["config"]["launch_plan"])
    except Exception as e: print(n, "ERR", e)
PY
