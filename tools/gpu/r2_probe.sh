#!/bin/bash
# sustained-MFMA probe + kernel trace of the int8 bench (durations under hipGraph replay)
TAG=${1:-r2c}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_sustained $R/tools/probes/mfma_sustained.hip && timeout 300 /tmp/mfma_sustained > $R/gpurun_out/${TAG}_mfma_sustained.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_int8 -o t -- python $R/bench.py --config int8 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof_int8.json 2> $R/gpurun_out/${TAG}_prof_int8.err
cd $R
cp $(find gpurun_out/${TAG}_prof_int8 -name "t_kernel_stats.csv" | head -1) gpurun_out/${TAG}_int8_kernel_stats.csv
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$(find gpurun_out/${TAG}_prof_int8 -name 't_kernel_trace.csv' | head -1)")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last replay: take the last 1/25 of the dispatches
n = len(rows) // 26
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
with open("gpurun_out/${TAG}_int8_timeline.txt", "w") as f:
    prev_end = t0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%8.1f us  gap %5.1f  dur %6.1f  %-60s grid %s wg %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"]))
        prev_end = e
PY
find gpurun_out -name "t_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*.db" -delete
cat gpurun_out/${TAG}_mfma_sustained.txt; head -20 gpurun_out/${TAG}_int8_kernel_stats.csv | cut -c1-150
