"""Find reads of uninitialised device memory: torch.empty / empty_like / Tensor.new_empty are replaced by NaN- (float) or
0x7F-filled (integer) allocations, then one iteration runs eagerly and as a captured graph.  Anything that reads a
workspace before writing it now shows up as NaN in a loss, a parameter or a BatchNorm buffer instead of as a last-bit
run-to-run difference.

    python tools/poison_empty.py [2d|3d] [bf16|fp32]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0x7F)
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(0x7F7F7F7F)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

from chap_amd import synthetic as syn  # noqa: E402
from chap_amd.networks import DualDecoder, DualDecoder3d  # noqa: E402
from chap_amd.train import ChapStep  # noqa: E402


def report(tag, model, out):
    bad = [k for k, v in model.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
    losses = [float(l[2]) for l in out["mix_losses"]] + [float(out["vat_loss"])]
    print(tag, "losses", ["%.5f" % v for v in losses], "non-finite tensors:", len(bad), bad[:6], flush=True)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "3d"
    dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    dev = torch.device("cuda", 0)
    torch.manual_seed(1337)
    if cfg == "3d":
        B, sp = 4, (112, 112, 80)
        m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(dtype)
        a = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1, num_classes=2)
        vol, lab = syn.synthetic_batch_3d(1337, B // 2, B // 2, *sp)
    else:
        B, sp = 24, (256, 256)
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(dtype)
        a = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1)
        vol, lab = syn.synthetic_batch(1337, B // 2, B // 2, *sp)
    vol, lab = vol.to(dev), lab.to(dev)
    step = ChapStep(m, a)
    step.iter_num = 4500
    out = step.step(vol, lab)
    torch.cuda.synchronize()
    report("eager ", m, out)
    step.capture(vol, lab, warmup=1)
    out = step.replay(vol, lab)
    torch.cuda.synchronize()
    report("replay", m, out)


if __name__ == "__main__":
    main()
