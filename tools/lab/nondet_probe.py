"""Which tensors differ between two identical 3D iterations run after a 2D iteration in the same process?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chap_amd import synthetic as syn
from chap_amd.networks import DualDecoder, DualDecoder3d
from chap_amd.train import ChapStep
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
extra = {}
for kv in sys.argv[2:]:
    k, v = kv.split("="); extra[k] = (v == "1")
def run2d():
    torch.manual_seed(1); m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).train().set_compute_dtype(torch.bfloat16)
    st = ChapStep(m, dict(batch_size=24, labeled_bs=12)); v, l = syn.synthetic_batch(1, 12, 12, 256, 256)
    st.capture(v.to(dev), l.to(dev), warmup=1); st.replay(v.to(dev), l.to(dev)); torch.cuda.synchronize()
def run3d():
    torch.manual_seed(1337); np.random.seed(1337)
    m = DualDecoder3d(n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True).to(dev).train().set_compute_dtype(torch.bfloat16)
    st = ChapStep(m, dict(dict(batch_size=4, labeled_bs=2, vat_iters=1, num_classes=2), **extra)); st.iter_num = 4500
    v, l = syn.synthetic_batch_3d(1337, 2, 2, 112, 112, 80); v, l = v.to(dev), l.to(dev)
    if mode == "graph":
        st.capture(v, l, warmup=1); out = st.replay(v, l)
    else:
        out = st.step(v, l, box_yx=(3, 5, 9))
    torch.cuda.synchronize()
    return {k: t.clone() for k, t in m.state_dict().items()}, [x.clone() for x in out["mix_losses"]] + [out["vat_loss"].clone()], st.opt.mom.clone()
run2d()
a, la, ma = run3d()
b, lb, mb = run3d()
print("losses equal:", [bool(torch.equal(x, y)) for x, y in zip(la, lb)])
bad = [(k, float((a[k].float() - b[k].float()).abs().max())) for k in a if not torch.equal(a[k], b[k])]
print(len(bad), "of", len(a), "tensors differ")
for k, d in bad[:40]:
    print("  %-60s %.3e" % (k, d))
