"""The CONCURRENT timeline of one untraced HIP-graph replay of the training iteration.

A rocprofv3 kernel trace all but serialises the graph (10.06 ms per 2D step against 6.8 ms untraced), so it shows kernels alone, not the schedule.
Here every launch behind the library's trampolines (csrc/launch.h; lab build with -DCHAP_TIMELINE: `make -C chap_amd/csrc -f Makefile.lab`) stamps
the 100 MHz s_memrealtime counter into its own words -- the first, middle and last block of the grid a (start, end) pair each, plain stores (a stamp
per block slowed the step by 26-42 %, see launch.h).  The pointer is baked into the captured graph node, so after a replay the buffer holds the start /
end of every kernel of THAT replay.  The directly launched kernels (losses, largest-CC, VAT helpers, SGD: ~5 % of the launches) are followed by a
one-thread marker kernel whose stamp bounds their end ("<name> (end marker)" events).

    CHAP_LIBPATH=tools/lab/libchap_hip_lab.so python tools/timeline_untraced.py [--config 2d|3d] [--replays 6] [--out gpurun_out/r04_timeline_untraced_2d.json]

Output (JSON): step time of the instrumented replay and of the same graph with the stamps switched off is printed by bench.py; here: per stream
(chain) the busy time, the idle gaps and which kernels ran beside; time with 0 / 1 / 2 / 3+ kernels in flight; per kernel name launches, summed
duration, exclusive time; the critical chain (the capture's origin stream) with its largest gaps and what was running on the other streams then."""
import argparse
import ctypes as C
import json
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from chap_amd import _lib as L                                  # noqa: E402
from chap_amd.networks import DualDecoder, DualDecoder3d        # noqa: E402
from chap_amd.synthetic import synthetic_batch, synthetic_batch_3d   # noqa: E402
from chap_amd.train import ChapStep                             # noqa: E402

TICK_US = 0.01          # s_memrealtime: 100 MHz


def entries(lib, lo, hi):
    lib.chap_timeline_entry.restype = C.c_int
    lib.chap_timeline_entry.argtypes = [C.c_long, C.c_char_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_long)]
    out = []
    name = C.create_string_buffer(128)
    g = (C.c_uint * 3)()
    st, fn, first = C.c_void_p(), C.c_void_p(), C.c_long()
    for i in range(lo, hi):
        rc = lib.chap_timeline_entry(i, name, 128, g, C.byref(st), C.byref(fn), C.byref(first))
        assert rc == 0
        out.append(dict(name=name.value.decode(), grid=[g[0], g[1], g[2]], stream=st.value or 0, fn=fn.value or 0, first=first.value))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2d", choices=["2d", "3d"])
    ap.add_argument("--replays", type=int, default=6)
    ap.add_argument("--vat-iters", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--steady", type=int, default=4, help="replays issued back to back before the stamps are read (the last one is what the buffer holds)")
    a = ap.parse_args()
    lib = L.lib()
    if not hasattr(lib, "chap_timeline_enable"):
        raise SystemExit("this library has no timeline stamps: build tools/lab/libchap_hip_lab.so (make -C chap_amd/csrc -f Makefile.lab) and set CHAP_LIBPATH")
    dev = torch.device("cuda")
    nslots = 1 << 18                    # (start, end) pairs: three per launch recorded
    buf = torch.zeros(nslots, 2, dtype=torch.int64, device=dev)
    lib.chap_timeline_enable.argtypes = [C.c_void_p, C.c_long]
    lib.chap_timeline_count.restype = C.c_long
    lib.chap_timeline_enable(C.c_void_p(buf.data_ptr()), nslots)
    torch.manual_seed(1337)
    np.random.seed(1337)
    if a.config == "2d":
        B, sp = 24, (256, 256)
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"})
        vol, lab = synthetic_batch(1337, B // 2, B - B // 2, *sp)
        args = dict(labeled_bs=B // 2, batch_size=B, vat_iters=a.vat_iters)
    else:
        B, sp = 4, (112, 112, 80)
        m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
        vol, lab = synthetic_batch_3d(1337, B // 2, B - B // 2, *sp)
        args = dict(labeled_bs=B // 2, batch_size=B, vat_iters=a.vat_iters, num_classes=2)
    m = m.to(dev).train().set_compute_dtype(torch.bfloat16)
    step = ChapStep(m, args)
    vol, lab = vol.to(dev), lab.to(dev)
    marks = []
    orig = step.device_step

    def wrapped(*aa, **kk):
        marks.append(lib.chap_timeline_count())
        return orig(*aa, **kk)

    step.device_step = wrapped
    step.capture(vol, lab, warmup=2)
    lo, hi = marks[-1], lib.chap_timeline_count()             # the slots of the captured iteration
    ents = entries(lib, lo, hi)
    origin = step._cap.cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        step.replay(vol, lab)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.replays):
        step.replay(vol, lab)
    e1.record()
    torch.cuda.synchronize()
    ms_stamped = e0.elapsed_time(e1) / a.replays
    # zero the words, then --steady (default) replays back to back and read: every replay overwrites the same words, so the buffer ends up holding
    # the LAST one -- a replay in steady state, whose nodes the host queued while the previous replay was still running.  (--steady 1 = one isolated
    # replay behind a synchronize: the host feeds the nodes while the GPU already executes, ~3 us per node, and the chains idle at their forks:
    # profiles/r04_timeline_untraced_2d_isolated.json -- a start-up effect, not the steady state the bench measures.)
    p_lo = ents[0]["first"]
    p_hi = ents[-1]["first"] + 3
    buf[p_lo:p_hi].zero_()
    torch.cuda.synchronize()
    for _ in range(a.steady):
        step.replay(vol, lab)
    torch.cuda.synchronize()
    raw = buf[p_lo:p_hi].cpu().numpy().astype(np.int64)
    t = np.zeros((len(ents), 2), dtype=np.int64)
    life = []
    for i, e in enumerate(ents):
        r = raw[e["first"] - p_lo:e["first"] - p_lo + 3]                 # first / middle / last block of the grid
        good = (r[:, 0] > 0) & (r[:, 1] >= r[:, 0])
        if good.any():
            t[i] = (r[good, 0].min(), r[good, 1].max())
            life.append(float(np.median(r[good, 1] - r[good, 0])) * TICK_US)
        else:
            life.append(0.0)
    ok = (t[:, 0] > 0) & (t[:, 1] >= t[:, 0])
    t0 = int(t[ok, 0].min())
    ev = []
    for e, (s, f), good, lf in zip(ents, t, ok, life):
        if good:
            ev.append(dict(e, start_us=(int(s) - t0) * TICK_US, end_us=(int(f) - t0) * TICK_US, median_block_life_us=round(lf, 2)))
    ev.sort(key=lambda e: e["start_us"])
    span = max(e["end_us"] for e in ev)
    streams = sorted({e["stream"] for e in ev}, key=lambda s: (s != origin, s))
    sname = {s: ("origin" if s == origin else "s%d" % i) for i, s in enumerate(streams)}
    # concurrency profile
    pts = sorted([(e["start_us"], 1) for e in ev] + [(e["end_us"], -1) for e in ev], key=lambda p: (p[0], p[1]))
    conc, live, last = defaultdict(float), 0, 0.0
    for tt, d in pts:
        conc[min(live, 3)] += tt - last
        last, live = tt, live + d
    # per name
    per = defaultdict(lambda: [0, 0.0])
    for e in ev:
        per[e["name"]][0] += 1
        per[e["name"]][1] += e["end_us"] - e["start_us"]
    # per stream: busy, gaps; for the origin chain: the largest gaps with the co-runners during the gap
    chains = {}
    for s in streams:
        es = [e for e in ev if e["stream"] == s]
        busy = sum(e["end_us"] - e["start_us"] for e in es)
        gaps = []
        for x, y in zip(es, es[1:]):
            g = y["start_us"] - x["end_us"]
            if g > 0:
                co = sorted({"%s@%s" % (o["name"], sname[o["stream"]]) for o in ev if o["stream"] != s and o["start_us"] < y["start_us"] and o["end_us"] > x["end_us"]})
                gaps.append(dict(after=x["name"], before=y["name"], at_us=round(x["end_us"], 2), gap_us=round(g, 2), co_running=co[:6]))
        gsum = sum(g["gap_us"] for g in gaps)
        chains[sname[s]] = dict(launches=len(es), first_us=round(es[0]["start_us"], 2), last_us=round(es[-1]["end_us"], 2), busy_us=round(busy, 1), gap_us=round(gsum, 1),
                                gaps_over_5us=len([g for g in gaps if g["gap_us"] > 5]), median_gap_us=round(float(np.median([g["gap_us"] for g in gaps])) if gaps else 0.0, 2),
                                largest_gaps=sorted(gaps, key=lambda g: -g["gap_us"])[:12])
    # who co-runs with the origin chain's kernels: for each origin kernel name, the share of its time with k other kernels in flight
    oc = defaultdict(lambda: [0.0, 0.0])
    others = [e for e in ev if e["stream"] != origin]
    for e in ev:
        if e["stream"] != origin:
            continue
        d = e["end_us"] - e["start_us"]
        ov = sum(max(0.0, min(e["end_us"], o["end_us"]) - max(e["start_us"], o["start_us"])) for o in others)
        oc[e["name"]][0] += d
        oc[e["name"]][1] += ov
    out = dict(config=a.config, vat_iters=a.vat_iters, replays_back_to_back=a.steady, launches_stamped=len(ev), launches_recorded=hi - lo, unstamped=int((~ok).sum()),
               ms_per_step_with_stamps=round(ms_stamped, 3), span_of_the_read_replay_us=round(span, 1),
               time_with_n_kernels_in_flight_us={str(k) + ("+" if k == 3 else ""): round(v, 1) for k, v in sorted(conc.items())},
               per_kernel=[dict(name=n, launches=c, total_us=round(d, 1), avg_us=round(d / c, 2)) for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:a.top]],
               chains=chains,
               origin_chain_overlap=[dict(name=n, total_us=round(d, 1), other_kernel_us_beside=round(o, 1)) for n, (d, o) in sorted(oc.items(), key=lambda kv: -kv[1][0])[:a.top]],
               events=[dict(name=e["name"], grid=e["grid"], stream=sname[e["stream"]], start_us=round(e["start_us"], 2), end_us=round(e["end_us"], 2),
                            median_block_life_us=e["median_block_life_us"]) for e in ev])
    path = a.out or os.path.join(ROOT, "gpurun_out", "r04_timeline_untraced_%s.json" % a.config)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f)
    print(json.dumps({k: v for k, v in out.items() if k not in ("events", "chains", "per_kernel", "origin_chain_overlap")}))
    for s, c in out["chains"].items():
        print(s, json.dumps({k: v for k, v in c.items() if k != "largest_gaps"}))


if __name__ == "__main__":
    main()
