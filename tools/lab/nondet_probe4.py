import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_parity_gates_gpu import _full_size_run
keep = []
extra = sys.argv[1:] 
for rnd in range(3):
    runs = [_full_size_run("3d", torch.float32 if i == 0 else torch.bfloat16) for i in range(3)]
    a, b = runs[1], runs[2]
    eq = [bool(torch.equal(x, y)) for x, y in zip(a[2], b[2])]
    bad = [k for k in a[0].state_dict() if not torch.equal(a[0].state_dict()[k], b[0].state_dict()[k])]
    print("round", rnd, "losses equal per step", eq, "max diff per step", (a[2] - b[2]).abs().max(1).values.tolist(), "| tensors differing:", len(bad), bad[:4], flush=True)
    if "keep" in extra:
        keep.append(runs)
    del runs, a, b
