"""Print the top rows of a rocprofv3 kernel_stats.csv (usage: kstats.py <dir-or-csv> [rows])."""
import csv, glob, sys
p = sys.argv[1]
f = p if p.endswith('.csv') else glob.glob(p + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6:.2f} ms over {len(rows)} kernels ({f})")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls={r['Calls']:>6} avg={float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:120]}")
