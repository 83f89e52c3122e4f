"""Host-side overhead around the graph replay: ChapStep.replay() (static input copies, box/weight/lr updates) vs the bare graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chap_amd.networks import DualDecoder
from chap_amd.train import ChapStep
from chap_amd import synthetic

dev = "cuda"
B = 24
m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(torch.bfloat16)
vol, lab = synthetic.synthetic_batch(1, B // 2, B // 2, 256, 256)
vol, lab = vol.to(dev), lab.to(dev)
step = ChapStep(m, dict(batch_size=B, labeled_bs=B // 2))
step.capture(vol, lab)
for name, fn in (("replay()", lambda: step.replay(vol, lab)), ("bare graph", lambda: step._graph.replay())):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print("%-12s %.3f ms/step" % (name, (time.perf_counter() - t) / 50 * 1e3), flush=True)
