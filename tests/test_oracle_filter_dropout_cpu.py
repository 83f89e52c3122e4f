"""The oracle's channel-perturbation restatement (oracle/filter_dropout.py) against vectors produced by the imported
reference under scripted random draws (tests/golden/filter_dropout.npz, oracle/gen_golden.py: gen_filter_dropout)."""
import os

import numpy as np
import torch

from oracle import filter_dropout as ofd
from oracle import init as oinit

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "filter_dropout.npz"))
CASES = (("drop2d", False, False, 0), ("binom", False, True, 0), ("scores", True, False, 0),
         ("scores_comp0", True, True, 0), ("scores_comp1", True, True, 1))


def test_perform_dropout_masks_match_reference():
    feats, scores, uniforms = ofd.fd_inputs()
    level = [int(v) for v in G["level"]]
    B = feats[0].shape[0]
    for name, with_scores, comp, branch in CASES:
        f1, f2 = ofd.perform_dropout(feats, level, scores if with_scores else None, comp, uniforms, [branch] * 5)
        for idx, (a, b, f) in enumerate(zip(f1, f2, feats)):
            assert a.shape[0] == B + B // 2 and torch.equal(a[:B], f) and torch.equal(b[:B], f)
            unlab = f[B // 2:]
            for tag, t in (("m1", a), ("m2", b)):
                want = torch.from_numpy(G["%s_L%d_%s" % (name, idx, tag)])
                got = t[B:, :, 0, 0] / unlab[:, :, 0, 0]
                assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (name, idx, tag)
                if idx not in level:
                    assert torch.equal(t[B:], unlab)


def test_masks_have_the_documented_structure():
    feats, scores, uniforms = ofd.fd_inputs()
    g = {k: G[k] for k in G.files}
    assert set(np.unique(np.round(g["drop2d_L0_m1"], 4))) <= {0.0, 2.0}
    assert np.allclose(g["binom_L1_m1"] + g["binom_L1_m2"], 2.0, atol=1e-5)
    assert set(np.unique(np.round(g["scores_L2_m1"], 4))) <= {0.0, 2.0}              # all-zero scores: Dropout2d fallback
    m = g["scores_L4_m1"]
    assert abs(m.mean() - 1.0) < 1e-5 and len(np.unique(np.round(m, 4))) == 2          # mask * numel / sum
    assert np.allclose(g["scores_L3_m1"], 1.0, atol=1e-6)   # level 3 not perturbed


def test_dropout_forward_matches_reference():
    _, scores, uniforms = ofd.fd_inputs()
    sd = oinit.dual_decoder_2d_state(int(G["fwd_state_seed"]))
    x = torch.from_numpy(G["fwd_x"])
    masks = oinit.drop_masks_2d(int(G["fwd_mask_seed"]), *x.shape[0:1], *x.shape[2:])
    with torch.no_grad():
        o1, o2 = ofd.dual_decoder_2d_dropout(sd, x, [0, 1, 2, 3, 4], scores, True, uniforms, [1] * 5, train=True, drop=masks)
    assert o1.shape[0] == 6
    for o, key in ((o1, "fwd_logits1"), (o2, "fwd_logits2")):
        want = torch.from_numpy(G[key])
        assert float((o - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_dropout_backward_matches_reference():
    """Gradients through the perturbed pass (masks are constants, the pooled activation is detached, FilterDropout.py:76):
    eval mode in fp32 and train mode in fp64 against the reference's autograd."""
    _, scores, uniforms = ofd.fd_inputs()
    x0 = torch.from_numpy(G["fwd_x"])
    masks = oinit.drop_masks_2d(int(G["fwd_mask_seed"]), x0.shape[0], *x0.shape[2:])
    for tag, train, dtype, tol in (("eval", False, torch.float32, 2e-4), ("train64", True, torch.float64, 1e-5)):
        sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in oinit.dual_decoder_2d_state(int(G["fwd_state_seed"])).items()}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        x = x0.clone().to(dtype).requires_grad_(True)
        o1, o2 = ofd.dual_decoder_2d_dropout(sd, x, [0, 1, 2, 3, 4], scores, True, uniforms, [1] * 5, train=train, drop=masks)
        g = torch.Generator().manual_seed(int(G["bwd_cot_seed"]))
        loss = sum((o * torch.randn(o.shape, generator=g).to(dtype)).sum() for o in (o1, o2))
        loss.backward()
        assert _relerr(o1.detach().numpy(), G["bwd_%s_logits1" % tag]) < tol
        assert _relerr(x.grad.numpy(), G["bwd_%s_dx" % tag]) < 10 * tol
        for i, n in enumerate(G["bwd_pick_names"]):
            assert _relerr(sd[str(n)].grad.numpy(), G["bwd_%s_grad_pick%d" % (tag, i)]) < 10 * tol, (tag, n)
