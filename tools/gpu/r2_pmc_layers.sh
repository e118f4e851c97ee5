#!/bin/bash
# PMC passes over isolated layers / plans: tools/gpu/r2_pmc_layers.sh <tag> <layers> <plans>
TAG=$1; LAYERS=$2; PLANS=$3
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}_$n -o $n -- python $R/tools/layer_probe.py --layers $LAYERS --variants $PLANS --reps 3 > $R/gpurun_out/${TAG}_$n.log 2>&1
  echo "$n rc=$?" >> $R/gpurun_out/${TAG}_$n.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
cd $R
python - <<PY
import csv, collections
def load(name):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open("gpurun_out/${TAG}_%s/%s_counter_collection.csv" % (name, name))):
        if "igemm" not in r["Kernel_Name"]: continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:60], "grid": int(r["Grid_Size"]), "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(rows.values())
a, b = load("sq1"), load("sq2")
print("kernel,workgroups,dur_us,mfma_busy,clock_mhz,wait_any,wait_inst_any,wait_inst_lds,active,valu/mfma,salu/mfma,lds/mfma,vmem/mfma,lds_conflict_frac")
for x, y in zip(a, b):
    gui = x["GRBM_GUI_ACTIVE"] / 8; wc = max(x["SQ_WAVE_CYCLES"], 1); mf = max(y["SQ_INSTS_MFMA"], 1)
    print(f'{x["k"]},{x["grid"]//256},{x["t"]:.1f},{x["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024):.3f},{gui/x["t"]:.0f},{x["SQ_WAIT_ANY"]/wc:.3f},{x["SQ_WAIT_INST_ANY"]/wc:.3f},{x["SQ_WAIT_INST_LDS"]/wc:.3f},{x["SQ_ACTIVE_INST_ANY"]/wc:.3f},'
          f'{y["SQ_INSTS_VALU"]/mf:.2f},{y["SQ_INSTS_SALU"]/mf:.2f},{y["SQ_INSTS_LDS"]/mf:.2f},{y["SQ_INSTS_VMEM_RD"]/mf:.2f},{y["SQ_LDS_BANK_CONFLICT"]/max(y["SQ_LDS_IDX_ACTIVE"],1):.3f}')
PY
find gpurun_out -name "*kernel_trace*" -size +1M -delete; find gpurun_out -name "*.db" -delete
