"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass (separate from any trace
pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"): usage  pmc_mfma.py <dir> [name-substring ...]  -> JSON lines.

    util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)

rocprofv3 reports both counters summed over their instances: GRBM_GUI_ACTIVE over the 8 XCDs (so / 8 = the dispatch's
cycles), SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs (it counts cycles: 16 per v_mfma_f32_16x16x32_bf16); 1024 = 256 CUs x 4
SIMDs.  The guide's caveat applies: GUI_ACTIVE / 8 reads high on dispatches shorter than ~0.3 ms (it includes the
dispatch's ramp), so for the 10-100 us kernels here the utilisation is a LOWER bound; the flop-based fraction in the shape
table (algorithmic FLOPs / measured time / 2.5 PFLOP/s) is the other side."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    if subs and not any(s in k for s in subs):
        continue
    busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
    if busy == 0:
        continue
    print(json.dumps({"kernel": k.replace("void ", "").split("(")[0], "grid": int(grid), "dispatches": len(cs["GRBM_GUI_ACTIVE"]),
                      "SQ_VALU_MFMA_BUSY_CYCLES": round(busy), "GRBM_GUI_ACTIVE_sum8": round(gui), "mfma_util": round(busy / (gui / 8 * 1024), 4)}))
