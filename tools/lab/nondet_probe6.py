import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_parity_gates_gpu import _full_size_run
base = None
dummies = []
cfg = sys.argv[1] if len(sys.argv) > 1 else "3d"
for off in range(0, 36):
    m, st, l, _, _ = _full_size_run(cfg, torch.bfloat16, steps=2)
    ids = (st._side.cuda_stream & 0xffff, st._d2.cuda_stream & 0xffff, st._pre.cuda_stream & 0xffff, torch.cuda.graphs.graph.default_capture_stream.cuda_stream & 0xffff)
    if base is None:
        base = l
    print("run", off, "equal to run 0:", bool(torch.equal(l, base)), "maxdiff %.2e" % float((l - base).abs().max()), "streams side/d2/pre/capture:", ["%04x" % i for i in ids], flush=True)
    del m, st
    dummies.append(torch.cuda.Stream())      # shift the pool's round-robin by one more
