"""Where does the iteration's wall time go?  Graph-replay time of ChapStep with parts switched off
(modes: full | novat | noconc | nob = without pass B: the VAT chain alone | nofork = decoders back to back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chap_amd.networks import DualDecoder, DualDecoder3d
from chap_amd.train import ChapStep
from chap_amd import synthetic as ots

def run(cfg, mode):
    dev = "cuda"
    if cfg == "2d":
        B, sp = 24, (256, 256)
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(torch.bfloat16)
        vol, lab = ots.synthetic_batch(1, B // 2, B // 2, *sp)
        a = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1)
    else:
        B, sp = 4, (112, 112, 80)
        m = DualDecoder3d(1, 2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(torch.bfloat16)
        vol, lab = ots.synthetic_batch_3d(1, B // 2, B // 2, *sp)
        a = dict(batch_size=B, labeled_bs=B // 2, vat_iters=1, num_classes=2)
    if mode == "novat": a["adv_noise"] = False
    if mode == "noconc": a["concurrent"] = False
    step = ChapStep(m, a)
    if mode == "nob":
        step._phase_b_steps = lambda ctx: iter(())          # (a generator that ends at once: phase B issues nothing; losses = None)
    if mode == "nofork":
        import contextlib
        step._decoder_fork = lambda origin: contextlib.nullcontext()
    vol, lab = vol.to(dev), lab.to(dev)
    step.capture(vol, lab)
    for _ in range(3): step.replay(vol, lab)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step.replay(vol, lab)
    torch.cuda.synchronize()
    print("%s %-7s %.3f ms/step" % (cfg, mode, (time.perf_counter() - t) / 20 * 1e3), flush=True)

if __name__ == "__main__":
    for cfg in ("2d", "3d"):
        for mode in ("full", "novat", "noconc", "nob", "nofork"):
            run(cfg, mode)
