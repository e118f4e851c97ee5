#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from one rocprofv3 --pmc pass (counters of tools/gpu/r2_prof.sh `sq1`):
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs), GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs), SQ_WAVE_CYCLES,
    SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY (quad-cycles, summed over waves), SQ_BUSY_CYCLES.
mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); effective clock = GUI_ACTIVE / 8 / duration.
usage: pmc_mfma.py <counter_collection.csv> [min launches] > table.csv"""
import collections, csv, re, sys


def norm(name):
    m = re.search(r"(igemm_\w+<[^>]*>|\w+_kernel\b[^()]*|\w+)", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    return m.group(1).replace(" ", "") if m else name


disp = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        d = disp.setdefault(r["Dispatch_Id"], {"k": norm(r["Kernel_Name"]), "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    a = agg[d["k"]]
    a["n"] += 1
    for k, v in d.items():
        if k != "k":
            a[k] += v
print("kernel,launches,avg_us,share_of_gpu_time,mfma_busy,effective_clock_mhz,wait_any,wait_inst_any,active_inst_any")
tot = sum(a["t"] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
    gui = a["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(a["SQ_WAVE_CYCLES"], 1.0)
    if a["n"] < (int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        continue
    print(f"{k},{int(a['n'])},{a['t'] / a['n']:.2f},{a['t'] / tot:.4f},{a['SQ_VALU_MFMA_BUSY_CYCLES'] / max(gui * 1024, 1):.3f},{gui / max(a['t'], 1e-9):.0f},"
          f"{a['SQ_WAIT_ANY'] / wc:.3f},{a['SQ_WAIT_INST_ANY'] / wc:.3f},{a['SQ_ACTIVE_INST_ANY'] / wc:.3f}")
