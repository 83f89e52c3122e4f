import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from chap_amd.networks import DualDecoder
from chap_amd.train import ChapStep
from chap_amd import synthetic as ots
dev = "cuda"
B = 24
m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(torch.bfloat16)
vol, lab = ots.synthetic_batch(1, B // 2, B // 2, 256, 256)
step = ChapStep(m, dict(batch_size=B, labeled_bs=B // 2, vat_iters=1))
vol, lab = vol.to(dev), lab.to(dev)
step.capture(vol, lab)
for _ in range(3): step.replay(vol, lab)
torch.cuda.synchronize()
t0 = time.perf_counter(); ts = []
for _ in range(20):
    a = time.perf_counter(); step.replay(vol, lab); ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop %.2f ms, until GPU done %.2f ms; per-replay host time: min %.3f max %.3f mean %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, min(ts) * 1e3, max(ts) * 1e3, sum(ts) / len(ts) * 1e3))
print(["%.2f" % (x * 1e3) for x in ts])
