"""Summarise rocprofv3 --pmc CSV output: per kernel, average counter value per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "conv_fwd_kernel" not in k and len(sys.argv) < 3:
        continue
    for c, v in cs.items():
        print("%-90s %-12s n=%4d avg=%14.1f" % (k[:90], c, len(v), sum(v) / len(v)))
