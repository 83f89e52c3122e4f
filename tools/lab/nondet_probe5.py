import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from chap_amd import ops
from oracle import train_step as ots
from tests.test_parity_gates_gpu import _full_size_run
DEV = "cuda"
parts = set(a for a in sys.argv[1:] if "=" not in a)
import json
import chap_amd.train as TR
_extra = {a.split("=")[0]: json.loads(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
_init = TR.ChapStep.__init__
def _patched(self, model, args=None, **kw):
    _init(self, model, dict(args or {}, **_extra), **kw)
TR.ChapStep.__init__ = _patched
for rnd in range(2):
    m32, s32, l32, (vol, lab), st32 = _full_size_run("3d", torch.float32)
    m16, s16, l16, _, st16 = _full_size_run("3d", torch.bfloat16)
    m16b, s16b, l16b, _, _ = _full_size_run("3d", torch.bfloat16)
    print("round", rnd, "equal:", bool(torch.equal(l16, l16b)), (l16 - l16b).abs().max(1).values.tolist(), flush=True)
    if "fwd" in parts:
        with torch.no_grad():
            pre1, pre2 = m32(vol[vol.shape[0] // 2:], update_stats=False)
            _, _, a1, a2, know = ops.pseudo_block(pre1, pre2)
    else:
        a1 = (torch.rand(2, 112, 112, 80, device=DEV) > 0.5).long(); a2 = a1.clone(); know = torch.rand(2, 112, 112, 80, device=DEV)
    if "lcc" in parts:
        got = ops.largest_cc(a1, 2)
        if "lccref" in parts:
            assert torch.equal(got.cpu(), ots.largest_cc(a1.cpu(), 2))
        ops.largest_cc(got, 2)
    if "box" in parts:
        box = torch.tensor([3, 5, 9, 74, 74, 53], dtype=torch.int32, device=DEV)
        mask = torch.empty((1,) + tuple(vol.shape[2:]), dtype=torch.int64, device=DEV)
        ops.box_mask(mask, box)
        mixed = torch.empty_like(vol[:1]); ops.box_mix(vol[:1], vol[1:2], mixed, box)
    if "dm" in parts:
        ops.diff_mask(a1, a2, know, 4, 0.1)
    torch.cuda.synchronize()
