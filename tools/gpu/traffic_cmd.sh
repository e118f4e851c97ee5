#!/bin/bash
# HBM traffic per kernel of ANY harness command: two PMC passes (FETCH_SIZE, WRITE_SIZE) plus one plain kernel-trace pass
# for the durations.  Usage: tools/gpu/traffic_cmd.sh <tag> <command ...>      (counters and timing in separate runs)
TAG=$1; shift
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_$c -o t -- "$@" > $R/gpurun_out/${TAG}_$c.log 2>&1)
  echo "$c rc=$?"
done
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_time -o t -- "$@" > $R/gpurun_out/${TAG}_time.log 2>&1)
cd $R
python tools/pmc_traffic.py gpurun_out/${TAG}_FETCH_SIZE/t_counter_collection.csv gpurun_out/${TAG}_WRITE_SIZE/t_counter_collection.csv > gpurun_out/${TAG}_hbm_traffic_per_kernel.json
python - <<PY
import csv, json, re
t = json.load(open("gpurun_out/${TAG}_hbm_traffic_per_kernel.json"))["kernels"]
def norm(name):
    m = re.search(r"(igemm_\w+<[^>]*>|\w+_kernel\b[^()]*|\w+)", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    return m.group(1).replace(" ", "") if m else name
rows = []
for r in csv.DictReader(open("gpurun_out/${TAG}_time/t_kernel_stats.csv")):
    k = norm(r["Name"])
    if k in t:
        b = t[k]["hbm_read_bytes_per_launch"] + t[k]["hbm_write_bytes_per_launch"]
        us = float(r["AverageNs"]) / 1e3
        rows.append((float(r["Percentage"]), k, int(r["Calls"]), us, b, b / us / 1e6))
print("%-52s %6s %9s %12s %9s %6s" % ("kernel", "calls", "avg us", "HBM B/launch", "TB/s", "time%"))
for pct, k, calls, us, b, tbs in sorted(rows, reverse=True)[:16]:
    print("%-52s %6d %9.1f %12d %9.2f %6.1f" % (k[:52], calls, us, b, tbs, pct))
PY
find gpurun_out -name "t_kernel_trace.csv" -size +3M -delete; find gpurun_out -name "t_counter_collection.csv" -size +8M -delete
