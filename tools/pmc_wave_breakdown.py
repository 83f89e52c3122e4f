"""Where the wave cycles of a kernel go, from ONE `rocprofv3 --pmc` pass (no trace flags) with the eight SQ counters
    SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
(MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY = wave parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing;
the three are disjoint and sum to ~WAVE_CYCLES; all four count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles).

    python3 tools/pmc_wave_breakdown.py DIR [--last N] [kernel-substring ...]   -> JSON lines, one per (kernel, grid)

--last N: only the last N dispatches of each kernel (the timing loop of tools/shape_table.py --eager --reps N)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import short_name  # noqa: E402

args = sys.argv[1:]
d = args.pop(0)
last = 0
if args and args[0] == "--last":
    last = int(args[1]); args = args[2:]
subs = args
rows = collections.defaultdict(lambda: collections.defaultdict(dict))       # (kernel, grid) -> dispatch id -> counter -> value
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(short_name(r["Kernel_Name"]), r["Grid_Size"])][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for (k, grid), disp in sorted(rows.items()):
    if subs and not any(s in k for s in subs):
        continue
    ids = sorted(disp)
    if last:
        ids = ids[-last:]
    n = len(ids)
    avg = collections.defaultdict(float)
    for i in ids:
        for c, v in disp[i].items():
            avg[c] += v / n
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    out = {"kernel": k, "grid_size": int(grid), "dispatches": n, "wave_quad_cycles": round(wc)}
    for c, key in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_ACTIVE_INST_VALU", "issuing_valu"),
                   ("SQ_ACTIVE_INST_LDS", "issuing_lds"), ("SQ_WAIT_INST_LDS", "issue_stall_lds")):
        if c in avg:
            out[key + "_frac"] = round(avg[c] / wc, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        out["mfma_busy_cycles_per_wave_cycle"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4)     # cycles / (quad-cycles x 4)
    print(json.dumps(out))
