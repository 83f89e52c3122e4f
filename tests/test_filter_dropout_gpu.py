"""Channel-level perturbation on the GPU against the oracle (oracle/filter_dropout.py, pinned to the
reference by tests/golden/filter_dropout.npz): the mask kernels and DualDecoder.forward(dropout=True)."""
import os

import numpy as np
import pytest
import torch

from oracle import filter_dropout as ofd
from oracle import init as oinit

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "filter_dropout.npz"))
CASES = (("drop2d", False, False, 0), ("binom", False, True, 0), ("scores", True, False, 0),
         ("scores_comp0", True, True, 0), ("scores_comp1", True, True, 1))


def _cl(x, dtype=torch.float32):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


@pytest.mark.parametrize("name,with_scores,comp,branch", CASES)
def test_channel_drop_masks_match_golden(name, with_scores, comp, branch):
    from chap_amd import ops
    feats, scores, uniforms = ofd.fd_inputs()
    B = feats[0].shape[0]
    U = B // 2
    for idx, f in enumerate(feats):
        Cc = f.shape[1]
        u1, u2 = (u.to(DEV) for u in uniforms[idx])
        mul1, mul2 = torch.empty(B + U, Cc, device=DEV), torch.empty(B + U, Cc, device=DEV)
        if with_scores:
            pooled = ops.sample_channel_sum(ops.Lazy(_cl(f[U:]).to(DEV)), nchunk=4)
            probs = torch.empty(U, Cc, device=DEV)
            ops.channel_drop(mul1, mul2, u1, u2, B, "scores", pool_partial=pooled, npix=f.shape[2] * f.shape[3],
                             grad_sim=scores[idx].to(DEV), comp=comp, branch=branch, probs_out=probs)
            if not bool(torch.all(scores[idx].eq(0))):
                want_p = ofd.drop_probs(scores[idx], f[U:].mean(dim=(2, 3)))
                assert float((probs.cpu() - want_p).abs().max()) < 2e-5
        else:
            ops.channel_drop(mul1, mul2, u1, u2, B, "comp_binomial" if comp else "dropout2d")
        assert torch.equal(mul1[:B].cpu(), torch.ones(B, Cc)) and torch.equal(mul2[:B].cpu(), torch.ones(B, Cc))
        for tag, m in (("m1", mul1), ("m2", mul2)):
            want = torch.from_numpy(G["%s_L%d_%s" % (name, idx, tag)])
            # the golden file perturbs levels 0, 1, 2, 4; level 3 has all-ones there, the kernel is still exercised
            if idx == 3:
                continue
            assert torch.allclose(m[B:].cpu(), want, rtol=1e-5, atol=1e-6), (name, idx, tag)


def test_sample_channel_sum_lazy_bf16():
    from chap_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 32, 20, 12, generator=g)
    sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        raw = _cl(x, dtype).to(DEV)
        part = ops.sample_channel_sum(ops.Lazy(raw, sc.to(DEV), sh.to(DEV), True, 0.01), nchunk=7)
        got = part.sum(1).cpu() / (20 * 12)
        xr = raw.float().cpu().permute(0, 3, 1, 2)
        want = torch.nn.functional.leaky_relu(xr * sc[None, :, None, None] + sh[None, :, None, None], 0.01).mean(dim=(2, 3))
        assert float((got - want).abs().max()) < tol


def test_dualdecoder_dropout_forward_matches_golden_and_oracle():
    from chap_amd.networks import DualDecoder
    _, scores, uniforms = ofd.fd_inputs()
    sd = oinit.dual_decoder_2d_state(int(G["fwd_state_seed"]))
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"})
    m.load_state_dict(sd, strict=True)
    m.to(DEV).train()
    x = torch.from_numpy(G["fwd_x"])
    masks = oinit.drop_masks_2d(int(G["fwd_mask_seed"]), x.shape[0], *x.shape[2:])
    dm = {k: _cl(v, torch.uint8).unsqueeze(1).to(DEV) for k, v in masks.items()}
    with torch.no_grad():
        o1, o2 = m(x.to(DEV), False, True, [0, 1, 2, 3, 4], [s.to(DEV) for s in scores], True, drop_masks=dm,
                   drop_uniforms=uniforms, drop_branches=[1] * 5)
    assert o1.shape == (6, 4, 32, 32)
    for o, key in ((o1, "fwd_logits1"), (o2, "fwd_logits2")):
        want = torch.from_numpy(G[key])
        assert float((o.cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    # a second configuration straight against the oracle: plain Dropout2d pairs on levels 1 and 3 only
    with torch.no_grad():
        a1, a2 = m(x.to(DEV), False, True, [1, 3], None, False, drop_masks=dm, drop_uniforms=uniforms, update_stats=False)
        w1, w2 = ofd.dual_decoder_2d_dropout({k: v.cpu() for k, v in m.state_dict().items()}, x, [1, 3], None, False, uniforms,
                                             train=True, drop=masks, update_stats=False)
    for o, want in ((a1, w1), (a2, w2)):
        assert float((o.cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def _cos(a, b):
    a, b = a.double().flatten().cpu(), torch.as_tensor(b).double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def _rel(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_backward_matches_golden(dtype):
    """Backward through the perturbed pass: the decoders' backward over B + U samples with chan_mul sources, folded back
    onto the encoder's B samples (chap_fold_perturbed) -- input and parameter gradients against the reference's autograd."""
    from chap_amd.networks import DualDecoder
    _, scores, uniforms = ofd.fd_inputs()
    x0 = torch.from_numpy(G["fwd_x"])
    masks = oinit.drop_masks_2d(int(G["fwd_mask_seed"]), x0.shape[0], *x0.shape[2:])
    dm = {k: _cl(v, torch.uint8).unsqueeze(1).to(DEV) for k, v in masks.items()}
    for tag, train in (("eval", False), ("train64", True)):
        m = DualDecoder(1, 4, {"decoder_type": "mcnet"})
        m.load_state_dict(oinit.dual_decoder_2d_state(int(G["fwd_state_seed"])), strict=True)
        m.to(DEV).set_compute_dtype(dtype).train(train)
        x = x0.clone().to(DEV).requires_grad_(True)
        o1, o2 = m(x, False, True, [0, 1, 2, 3, 4], [s.to(DEV) for s in scores], True, drop_masks=dm,
                   drop_uniforms=uniforms, drop_branches=[1] * 5)
        assert o1.shape[0] == 6
        g = torch.Generator().manual_seed(int(G["bwd_cot_seed"]))
        cots = [torch.randn(o.shape, generator=g).to(DEV) for o in (o1, o2)]
        torch.autograd.backward([o1, o2], cots)
        torch.cuda.synchronize()
        grads = dict(m.named_parameters())
        if dtype == torch.float32:
            assert _rel(o1.detach(), G["bwd_%s_logits1" % tag]) < 1e-4
        if dtype == torch.float32 and not train:
            assert _rel(x.grad, G["bwd_eval_dx"]) < 2e-3
            for i, n in enumerate(G["bwd_pick_names"]):
                assert _rel(grads[str(n)].grad, G["bwd_eval_grad_pick%d" % i]) < 2e-3, n
            got = np.array([float(p.grad.double().abs().sum()) for _, p in m.named_parameters()])
            np.testing.assert_allclose(got, G["bwd_eval_grad_checks"][:, 1], rtol=5e-3, atol=1e-4)
        elif dtype == torch.float32:
            # train-mode BN over 6 x 2 x 2 values at the bottleneck: fp32 sits ~1e-2 from the fp64 run (see test_net2d_gpu)
            assert _cos(x.grad, G["bwd_train64_dx"]) > 0.999 and _rel(x.grad, G["bwd_train64_dx"]) < 1e-1
            for i, n in enumerate(G["bwd_pick_names"]):
                want = G["bwd_train64_grad_pick%d" % i]
                if np.abs(want).max() < 1e-9:
                    continue
                assert _cos(grads[str(n)].grad, want) > 0.999, n
        else:
            assert _cos(x.grad, G["bwd_%s_dx" % tag]) > (0.9 if not train else 0.6)


def test_dropout_forward_needs_even_batch_and_device_rng():
    from chap_amd.networks import DualDecoder
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    x = torch.rand(2, 1, 32, 32, device=DEV)
    with torch.no_grad(), pytest.raises(ValueError):
        m(torch.rand(3, 1, 32, 32, device=DEV), False, True, [0], None, False)
    with torch.no_grad():                       # device RNG path: B + U outputs, finite
        o1, o2 = m(x, False, True, [0, 1, 2, 3, 4], None, True)
    assert o1.shape[0] == 3 and bool(torch.isfinite(o1).all()) and bool(torch.isfinite(o2).all())
    o1, o2 = m(x, False, True, [0, 1, 2, 3, 4], None, False)      # under autograd: parameter gradients arrive
    (o1.sum() + o2.sum()).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_gauss_probabilities_and_with_feat():
    """scores_dropoutV2(type='gauss') (FilterDropout.py:126-130) and dropout=True together with with_feat=True."""
    from chap_amd import ops
    from chap_amd.networks import DualDecoder
    feats, scores, uniforms = ofd.fd_inputs()
    f, B = feats[1], feats[1].shape[0]
    U, Cc = B // 2, f.shape[1]
    mul1, mul2 = torch.empty(B + U, Cc, device=DEV), torch.empty(B + U, Cc, device=DEV)
    probs = torch.empty(U, Cc, device=DEV)
    pooled = ops.sample_channel_sum(ops.Lazy(_cl(f[U:]).to(DEV)), nchunk=3)
    ops.channel_drop(mul1, mul2, uniforms[1][0].to(DEV), uniforms[1][1].to(DEV), B, "scores", pool_partial=pooled,
                     npix=f.shape[2] * f.shape[3], grad_sim=scores[1].to(DEV), prob_kind="gauss", probs_out=probs)
    want = ofd.drop_probs(scores[1], f[U:].mean(dim=(2, 3)), "gauss")
    assert float((probs.cpu() - want).abs().max()) < 2e-5
    m1, m2 = ofd.drop_based_on_prob(want, False, uniforms[1][0], uniforms[1][1])
    assert torch.allclose(mul1[B:].cpu(), m1[..., 0, 0], rtol=1e-5) and torch.allclose(mul2[B:].cpu(), m2[..., 0, 0], rtol=1e-5)
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).eval()
    x = torch.rand(2, 1, 32, 32, device=DEV)
    with torch.no_grad():
        o1, o2, fe = m(x, True, True, [4], None, False)
        p1, p2, fe0 = m(x, True)
    assert o1.shape[0] == 3 and len(fe) == 5 and fe[0].shape == (2, 16, 32, 32)
    assert torch.equal(fe[0], fe0[0])
    # eval mode: the first B samples of the perturbed pass see the same features and the same (running) BN statistics
    assert float((o1[:2] - p1).abs().max()) < 1e-5 and float((o2[:2] - p2).abs().max()) < 1e-5
