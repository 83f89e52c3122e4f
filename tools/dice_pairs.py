"""Is the bf16 throughput mode BIASED against fp32, or merely as noisy as fp32 is against itself?  

For S seeds: train BASELINE config 1 (2D DualDecoder, B = 24 = 12 + 12, 256 x 256; synthetic fixed-seed data, graph replay, `--iters`
iterations with the poly LR running to zero) once in fp32 and once in bf16 FROM THE SAME SEED -- identical initial weights, data order,
BCP boxes, dropout masks and VAT noise, so the only difference inside a pair is the arithmetic -- and evaluate the reference's inference
recipe (logit ensemble + arg-max, test_2D_fully.py:69-75) on 24 held-out slices.  Reported: mean +- s.e.m. of the mean foreground Dice per
mode over the seeds, the seed-to-seed standard deviation of each mode, and the PAIRED difference d_s = Dice_bf16(s) - Dice_fp32(s): mean,
standard deviation, s.e.m. and t = mean / s.e.m. (|t| < 2.78 = no bias detectable at the 5 % level with 5 pairs; 2.57 with 6).

    python tools/dice_pairs.py [--seeds 6] [--iters 1500] [--size 256] [--batch 24] [--out gpurun_out/r04_dice_pairs.json]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from chap_amd.networks import DualDecoder                       # noqa: E402
from chap_amd.synthetic import synthetic_batch                  # noqa: E402
from chap_amd.train import ChapStep                             # noqa: E402

DEV = "cuda"


def dice_per_class(pred, gt, n_classes=4):
    out = []
    for c in range(1, n_classes):
        p, g = pred == c, gt == c
        den = p.sum() + g.sum()
        out.append(2.0 * float((p & g).sum()) / float(den) if den > 0 else 1.0)
    return np.array(out)


def train_and_dice(dtype, seed, B, H, W, iters, lr=0.05):
    lbs = B // 2
    pool = [synthetic_batch(2000 + 100 * seed + i, lbs, B - lbs, H, W) for i in range(16)]
    pool = [(v.to(DEV), l.to(DEV)) for v, l in pool]
    val, gt = synthetic_batch(4242, 24, 0, H, W)
    torch.manual_seed(1337 + seed)                    # initial weights
    np.random.seed(1337 + seed)                       # BCP boxes
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train().set_compute_dtype(dtype)
    torch.manual_seed(7000 + seed)                    # the model's device RNG (dropout masks, VAT noise) is created from torch's seed at first use
    step = ChapStep(m, dict(labeled_bs=lbs, batch_size=B, base_lr=lr, max_iterations=iters))
    step.capture(*pool[0], warmup=1)
    t0 = time.time()
    for it in range(iters):
        step.replay(*pool[it % len(pool)])
    torch.cuda.synchronize()
    secs = time.time() - t0
    m.eval()
    with torch.no_grad():
        o1, o2 = m(val.to(DEV))
    pred = torch.argmax(torch.softmax((o1 + o2) / 2.0, dim=1), dim=1).cpu().numpy()
    return dice_per_class(pred, gt.numpy()), secs


def stats(x):
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    sd = float(x.std(ddof=1)) if n > 1 else 0.0
    return {"mean": float(x.mean()), "sd": sd, "sem": sd / math.sqrt(n) if n > 1 else 0.0, "n": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_dice_pairs.json"))
    a = ap.parse_args()
    runs = []
    for s in range(a.seeds):
        d32, t32 = train_and_dice(torch.float32, s, a.batch, a.size, a.size, a.iters)
        d16, t16 = train_and_dice(torch.bfloat16, s, a.batch, a.size, a.size, a.iters)
        r = {"seed": s, "dice_fp32": d32.round(5).tolist(), "dice_bf16": d16.round(5).tolist(), "mean_fp32": float(d32.mean()), "mean_bf16": float(d16.mean()),
             "paired_diff": float(d16.mean() - d32.mean()), "train_seconds": {"fp32": round(t32, 1), "bf16": round(t16, 1)}}
        print(json.dumps(r), flush=True)
        runs.append(r)
    f32, b16, d = stats([r["mean_fp32"] for r in runs]), stats([r["mean_bf16"] for r in runs]), stats([r["paired_diff"] for r in runs])
    d["t"] = d["mean"] / d["sem"] if d["sem"] > 0 else 0.0
    tcrit = {2: 12.71, 3: 4.30, 4: 3.18, 5: 2.78, 6: 2.57, 7: 2.45, 8: 2.36}.get(a.seeds, 2.3)
    out = {"config": {"size": a.size, "batch": a.batch, "iterations": a.iters, "seeds": a.seeds, "lr": 0.05, "data": "synthetic fixed-seed"},
           "fp32": f32, "bf16": b16, "paired_diff_bf16_minus_fp32": d, "t_critical_5pct": tcrit,
           "bias_detected": bool(abs(d["t"]) > tcrit), "both_learn": bool(min(r["mean_fp32"] for r in runs) > 0.7 and min(r["mean_bf16"] for r in runs) > 0.7),
           "runs": runs}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "runs"}))


if __name__ == "__main__":
    main()
