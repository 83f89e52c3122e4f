"""Join tools/shape_table.py's plan (--trace-plan) with the rocprofv3 --kernel-trace CSV of the same run: per
(op, shape) the kernel INSTANCE names and their trace durations.

    python tools/shape_join.py <shape_table.csv> <plan.json> <dir-with-*kernel_trace.csv> <out.csv>

The timing loop of shape i sits between the (2i)-th and (2i+1)-th `copy_kernel` marker launch; inside it the kernels are
grouped by name (a wgrad launch = wgrad_kernel + wgrad_reduce_kernel, act_bwd = reduce + apply, ...)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import short_name  # noqa: E402

table, plan_f, trace_dir, out = sys.argv[1:5]
rows = list(csv.DictReader(open(table)))
plan = json.load(open(plan_f))
tf = glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True)[0]
tr = list(csv.DictReader(open(tf)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(tr) if r["Kernel_Name"].startswith("copy_kernel")]
assert len(marks) >= 2 * len(plan), (len(marks), len(plan))
marks = marks[-2 * len(plan):]
by_key = {(r["op"], r["shape"]): r for r in rows}
MFMA = {"bf16": 2500.0e12, "fp32": 157.3e12}
outrows = []
for i, pl in enumerate(plan):
    seg = tr[marks[2 * i] + 1:marks[2 * i + 1]]
    per = collections.OrderedDict()
    for r in seg:
        per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # the graph replay holds `reps` launches of each kernel of the op
    kern = []
    tot_us = 0.0
    for name, ds in per.items():
        if len(ds) < pl["reps"]:
            continue                      # memset nodes etc.
        ds = ds[-pl["reps"] * (len(ds) // pl["reps"]):]
        us = sum(ds) / 1e3 / pl["reps"]
        tot_us += us
        kern.append("%s = %.2f us" % (short_name(name), us))
    r = dict(by_key[(pl["op"], pl["shape"])])
    r["trace_us"] = round(tot_us, 2)
    r["kernels"] = " | ".join(kern)
    by, fl = float(r["alg_MB"]) * 1e6, float(r["GFLOP"]) * 1e9
    if tot_us > 0:
        r["trace_hbm_frac"] = round(by / (tot_us * 1e-6) / 8000e9, 4)
        r["trace_TFLOPs"] = round(fl / tot_us / 1e6, 2)
    r["trace_us_per_iter"] = round(tot_us * int(r["launches_per_iter"]), 1)
    outrows.append(r)
outrows.sort(key=lambda r: -r["trace_us_per_iter"])
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(outrows[0].keys()))
    w.writeheader()
    w.writerows(outrows)
print("wrote %s (%d rows, %.2f ms of kernel time per iteration)" % (out, len(outrows), sum(r["trace_us_per_iter"] for r in outrows) / 1e3))
