"""Soak: many graph replays with changing inputs; losses and parameters must stay finite, the loss must fall."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chap_amd.networks import DualDecoder, DualDecoder3d
from chap_amd.train import ChapStep
from chap_amd import synthetic

dev = "cuda"
for cfg, iters in (("2d", 1500), ("3d", 300)):
    if cfg == "2d":
        B = 24
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(torch.bfloat16)
        pool = [synthetic.synthetic_batch(s, B // 2, B // 2, 256, 256) for s in range(4)]
        a = dict(batch_size=B, labeled_bs=B // 2)
    else:
        B = 4
        m = DualDecoder3d(1, 2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(torch.bfloat16)
        pool = [synthetic.synthetic_batch_3d(s, B // 2, B // 2, 112, 112, 80) for s in range(3)]
        a = dict(batch_size=B, labeled_bs=B // 2, num_classes=2)
    pool = [(v.to(dev), l.to(dev)) for v, l in pool]
    step = ChapStep(m, a)
    step.capture(*pool[0])
    t = time.perf_counter(); first = last = None
    for it in range(iters):
        out = step.replay(*pool[it % len(pool)])
        if it % (iters // 10) == 0 or it == iters - 1:
            tot = sum(float(x[2]) for x in out["mix_losses"]); vat = float(out["vat_loss"])
            assert tot == tot and abs(tot) < 1e4 and vat == vat, (it, tot, vat)
            first = tot if first is None else first; last = tot
            print("%s it %4d  bcp %.4f  vat %.5f  lr %.6f" % (cfg, it, tot, vat, step.opt.param_groups[0]["lr"]), flush=True)
    torch.cuda.synchronize()
    flat = m.flat_buffers()[0]
    assert torch.isfinite(flat).all() and torch.isfinite(step.opt.mom).all()
    print("%s: %d replays in %.1f s, loss %.4f -> %.4f, parameters finite" % (cfg, iters, time.perf_counter() - t, first, last), flush=True)
    assert last < first
