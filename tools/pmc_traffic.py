"""HBM-side traffic per kernel from two separate `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

    python tools/pmc_traffic.py [--last N] <fetch_dir> <write_dir> [name-substring ...]      -> JSON lines

traffic = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB; FETCH_SIZE counts a wide coalesced streaming read at half its bytes
on gfx950 -- MI355X_MICROARCH.md, HBM section; Infinity-Cache hits are included: this is L2-miss traffic).  Warm-up launches of
the timing loops are part of the average (same kernel, same shape)."""
import collections
import csv
import glob
import json
import re
import sys


LAST = 0
if "--last" in sys.argv:            # only the last N dispatches of every (kernel, grid): the timing loop of `shape_table.py --only ... --reps R` (3 + R launches)
    i = sys.argv.index("--last")
    LAST = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[(r["Kernel_Name"], r["Grid_Size"])].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    return {k: [x for _, x in sorted(v)][-LAST:] if LAST else [x for _, x in sorted(v)] for k, v in acc.items()}


import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import short_name as short  # noqa: E402


fd, wd = sys.argv[1], sys.argv[2]
subs = sys.argv[3:]
fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
for key in sorted(set(fe) | set(wr), key=lambda k: -sum(fe.get(k, [0]))):
    name, grid = key
    if subs and not any(s in name for s in subs):
        continue
    f = fe.get(key, [])
    w = wr.get(key, [])
    if not f or not w:
        continue
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    print(json.dumps({"kernel": short(name), "grid_size": int(grid), "dispatches": len(f), "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
                      "read_bytes": int(2 * fk * 1024), "write_bytes": int(wk * 1024), "traffic_bytes_per_launch": int((2 * fk + wk) * 1024)}))
