import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_parity_gates_gpu import _full_size_run
seq = sys.argv[1:] or ["2d:bf16", "3d:fp32", "3d:bf16", "3d:bf16"]
res = []
for s in seq:
    cfg, dt = s.split(":")
    m, st, l, _, _ = _full_size_run(cfg, torch.float32 if dt == "fp32" else torch.bfloat16)
    res.append((s, l, {k: v.clone() for k, v in m.state_dict().items()}))
    del m, st
a, b = res[-2], res[-1]
print(a[0], b[0], "losses bitwise equal per step:", [bool(torch.equal(x, y)) for x, y in zip(a[1], b[1])])
print((a[1] - b[1]).abs().max(1).values)
bad = [k for k in a[2] if not torch.equal(a[2][k], b[2][k])]
print(len(bad), "tensors differ", bad[:8])
