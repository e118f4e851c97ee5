#!/usr/bin/env python3
"""Summarise tools/gpu/pmc.sh output: the LAST `reps` dispatches of every igemm kernel per probed layer.
usage: pmc_summary.py gpurun_out/<tag> [reps=3]   (reads <tag>_sq1, _sq2, _fetch, _write)"""
import csv, sys, collections, re
tag = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3

def load(name):
    rows = collections.OrderedDict()
    with open(f"{tag}_{name}/{name}_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if "igemm" not in r["Kernel_Name"]: continue
            d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                                                       "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(rows.values())

def short(k):
    m = re.search(r"igemm_f32_(\w+?)(?:_kernel)?<([^>]*)>", k)
    return f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else k

sq1, sq2, fe, wr = load("sq1"), load("sq2"), load("fetch"), load("write")
assert len(sq1) == len(sq2) == len(fe) == len(wr), (len(sq1), len(sq2), len(fe), len(wr))
# group consecutive identical (kernel, grid) dispatches; keep groups as they come (layer order of the probe)
print("kernel,workgroups,dur_us,mfma_util,wait_any,wait_inst_any,active_inst,valu_per_mfma,salu_per_mfma,lds_per_mfma,vmem_rd_per_mfma,hbm_fetch_MB_x2,hbm_write_MB")
i = 0
while i < len(sq1):
    j = i
    while j < len(sq1) and sq1[j]["k"] == sq1[i]["k"] and sq1[j]["grid"] == sq1[i]["grid"]: j += 1
    sel = range(max(i, j - reps), j)
    n = len(sel)
    def avg(rows, key): return sum(rows[x].get(key, 0.0) for x in sel) / n
    a = {k: avg(sq1, k) for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE")}
    b = {k: avg(sq2, k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA")}
    dur = avg(sq1, "t")
    mf = max(b["SQ_INSTS_MFMA"], 1.0)
    util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8) if a["GRBM_GUI_ACTIVE"] else 0
    wc = max(a["SQ_WAVE_CYCLES"], 1.0)
    print(f"{short(sq1[i]['k'])},{sq1[i]['grid'] // sq1[i]['wg']},{dur:.1f},{util:.3f},{a['SQ_WAIT_ANY']/wc:.3f},{a['SQ_WAIT_INST_ANY']/wc:.3f},{a['SQ_ACTIVE_INST_ANY']/wc:.3f},"
          f"{b['SQ_INSTS_VALU']/mf:.2f},{b['SQ_INSTS_SALU']/mf:.2f},{b['SQ_INSTS_LDS']/mf:.2f},{b['SQ_INSTS_VMEM_RD']/mf:.2f},{2*avg(fe,'FETCH_SIZE')/1024:.1f},{avg(wr,'WRITE_SIZE')/1024:.1f}")
    i = j
