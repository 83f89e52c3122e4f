"""Concurrency timeline of the training iteration from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3
    python3 tools/timeline.py DIR [--iters 8] [--top 25]

The iteration is one HIP graph with forked streams, so the sum of the kernel durations says little about the step time.  This tool cuts the
trace into iterations (at the fused SGD kernel, the last node of the graph), and for the last --iters of them reports
  * wall time per iteration, the time with 0 / 1 / 2 / 3+ kernels in flight,
  * per kernel: launches, summed duration, EXCLUSIVE time (it is the only kernel in flight: step time it owns outright) and the idle time that
    follows it (the gap to the next kernel start while nothing else runs: dependent-launch latency on the critical path).
One JSON object on stdout."""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import short_name  # noqa: E402


def load(d):
    fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(d) else [d]
    rows = []
    for f in fs:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--delim", default="sgd", help="substring of the kernel that ends an iteration")
    a = ap.parse_args()
    rows = load(a.dir)
    ends = [e for s, e, n in rows if a.delim in n.lower()]
    if len(ends) < a.iters + 1:
        raise SystemExit("only %d delimiter kernels (%r) in the trace" % (len(ends), a.delim))
    t0, t1 = ends[-a.iters - 1], ends[-1]
    win = [(s, e, short_name(n)) for s, e, n in rows if s >= t0 and e <= t1]
    ev = []
    for i, (s, e, n) in enumerate(win):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()                                               # ends before starts at the same timestamp
    live = set()
    conc = defaultdict(int)
    excl = defaultdict(int)
    gap_after = defaultdict(int)
    gap_n = defaultdict(int)
    last_t, last_ended = t0, None
    for t, kind, i in ev:
        dt = t - last_t
        if dt > 0:
            conc[min(len(live), 3)] += dt
            if len(live) == 1:
                excl[win[next(iter(live))][2]] += dt
            elif len(live) == 0 and last_ended is not None:
                gap_after[win[last_ended][2]] += dt
                gap_n[win[last_ended][2]] += 1
        if kind == 1:
            live.add(i)
        else:
            live.discard(i)
            last_ended = i
        last_t = t
    tot = defaultdict(int)
    cnt = defaultdict(int)
    for s, e, n in win:
        tot[n] += e - s
        cnt[n] += 1
    it = float(a.iters)
    wall = (t1 - t0) / it
    out = {"iters": a.iters, "launches_per_iter": len(win) / it, "wall_us": wall / 1e3, "kernel_sum_us": sum(tot.values()) / it / 1e3,
           "in_flight_us": {("0 (idle)", "1", "2", "3+")[k]: conc[k] / it / 1e3 for k in range(4)}, "kernels": []}
    for n in sorted(tot, key=lambda n: -(excl[n] + gap_after[n]))[:a.top]:
        out["kernels"].append({"kernel": n, "launches": cnt[n] / it, "sum_us": round(tot[n] / it / 1e3, 1), "exclusive_us": round(excl[n] / it / 1e3, 1),
                               "idle_after_us": round(gap_after[n] / it / 1e3, 1), "idle_gaps": gap_n[n] / it})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
