"""rocprofv3 --stats kernel table -> the same table with a readable first column `Kernel` (tools/kname.py):
    python tools/kstats_short.py <in kernel_stats.csv> <out.csv>"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import short_name  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "Name"])
    for r in rows:
        w.writerow([short_name(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"], r["Name"]])
