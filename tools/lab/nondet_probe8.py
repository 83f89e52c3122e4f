"""vat_early: eager vs graph, with / without an unrelated eager forward in between.  Prints max |delta| of the per-step losses of two
identical runs (0.0 = bitwise equal).  usage: nondet_probe8.py [2d|3d] [key=json ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from chap_amd.networks import DualDecoder, DualDecoder3d
from chap_amd.train import ChapStep
from oracle import train_step as ots
DEV = "cuda"
cfg = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "2d"
extra = {a.split("=")[0]: json.loads(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}


def run(graph, steps=3):
    torch.manual_seed(1337); np.random.seed(1337)
    if cfg == "2d":
        B, sp = 24, (256, 256)
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"})
        args = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1)
        vol, lab = ots.synthetic_batch(1337, B // 2, B // 2, *sp)
    else:
        B, sp = 4, (112, 112, 80)
        m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
        args = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1, num_classes=2)
        vol, lab = ots.synthetic_batch_3d(1337, B // 2, B // 2, *sp)
    args.update(extra)
    m = m.to(DEV).train().set_compute_dtype(torch.bfloat16)
    step = ChapStep(m, args)
    step.iter_num = 4500
    vol, lab = vol.to(DEV), lab.to(DEV)
    if graph:
        step.capture(vol, lab, warmup=2)
    losses = []
    for i in range(steps):
        out = step.replay(vol, lab) if graph else step.step(vol, lab)
        losses.append(torch.cat([l[2:3] for l in out["mix_losses"]] + [out["vat_loss"]]).clone())
    torch.cuda.synchronize()
    return m, torch.stack(losses).cpu(), vol


for graph in (False, True):
    m1, l1, vol = run(graph)
    m2, l2, _ = run(graph)
    print("graph" if graph else "eager", "plain   : max delta", float((l1 - l2).abs().max()), flush=True)
    with torch.no_grad():
        m1(vol[vol.shape[0] // 2:], update_stats=False)          # the unrelated eager forward
    torch.cuda.synchronize()
    m3, l3, _ = run(graph)
    m4, l4, _ = run(graph)
    print("graph" if graph else "eager", "afterfwd: max delta vs first", float((l1 - l3).abs().max()), " 3 vs 4", float((l3 - l4).abs().max()), flush=True)
