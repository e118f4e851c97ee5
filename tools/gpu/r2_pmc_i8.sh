#!/bin/bash
# L2 / instruction counters of isolated int8 conv layers: tools/gpu/r2_pmc_i8.sh <tag> <layers>
TAG=$1; LAYERS=$2
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
run() { n=$1; shift
  LAYERS=$LAYERS timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}_$n -o $n -- python $R/tools/probe_int8_per_layer.py > $R/gpurun_out/${TAG}_$n.log 2>&1
  echo "$n rc=$?" >> $R/gpurun_out/${TAG}_$n.log; }
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
cd $R
python - <<PY
import csv, collections, os
for name in ("l2", "sq", "tcp"):
    path = "gpurun_out/${TAG}_%s/%s_counter_collection.csv" % (name, name)
    if not os.path.exists(path):
        print(name, "no csv"); os.system("tail -5 gpurun_out/${TAG}_%s.log" % name); continue
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "igemm_i8" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:40], int(r["Grid_Size"]))
        d = rows.setdefault(key, collections.defaultdict(float))
        d["n_" + r["Counter_Name"]] += 1
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["t"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for key, d in rows.items():
        cs = sorted(k for k in d if not k.startswith("n_") and k != "t")
        n = d["n_" + cs[0]]
        print(name, key, "launches", int(n), "avg_us %.1f" % (d["t"] / len(cs) / n), " ".join(f"{c}={d[c]/n:.0f}" for c in cs))
PY
find gpurun_out -name "*kernel_trace*" -size +1M -delete; find gpurun_out -name "*.db" -delete
