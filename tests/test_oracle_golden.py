"""The CPU oracle against the fixtures generated from the imported reference
(oracle/gen_golden.py).  CPU only; pins the oracle (task rule: parity is anchored
on the reference's own outputs)."""
import os

import numpy as np
import torch

from oracle import init as oinit
from oracle import nets


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _sd(state, grad=True, dtype=torch.float32):
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
    if grad:
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
    return sd


def _run(fn, sd, x, cot_seed, dtype=torch.float32, **kw):
    x = torch.from_numpy(x).clone().to(dtype).requires_grad_(True)
    outs = fn(sd, x, **kw)
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    g = torch.Generator().manual_seed(cot_seed)
    loss = sum((o * torch.randn(o.shape, generator=g).to(dtype)).sum() for o in outs)
    loss.backward()
    return outs, x.grad, loss


def _close(a, b, rtol=2e-4, atol=2e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _relerr(a, b):
    a = a.detach().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if np.abs(b).max() < 1e-9:      # mathematically-zero gradients (conv bias in front of train-mode BN)
        return np.abs(a).max()
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_dualdecoder2d_keys_and_eval(golden_dir):
    g = _load(golden_dir, "dualdecoder2d_64.npz")
    state = oinit.dual_decoder_2d_state(int(g["state_seed"]))
    assert list(state.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in state.values()] == list(g["shapes"])
    assert len(state) == 202
    sd = _sd(state)
    outs, dx, loss = _run(nets.dual_decoder_2d, sd, g["x"], int(g["cot_seed"]), train=False)
    assert _relerr(outs[0], g["eval_logits0"]) < 1e-5
    assert _relerr(outs[1], g["eval_logits1"]) < 1e-5
    assert _relerr(dx, g["eval_dx"]) < 1e-4
    for i, n in enumerate(g["grad_pick_names"]):
        assert _relerr(sd[str(n)].grad, g["eval_grad_pick%d" % i]) < 1e-4, n
    names = [str(n) for n in g["param_names"]]
    got = np.array([[float(sd[n].grad.double().sum()), float(sd[n].grad.double().abs().sum())] for n in names])
    np.testing.assert_allclose(got[:, 1], g["eval_grad_checks"][:, 1], rtol=2e-3, atol=1e-6)


def test_dualdecoder2d_train_injected_dropout(golden_dir):
    g = _load(golden_dir, "dualdecoder2d_64.npz")
    sd = _sd(oinit.dual_decoder_2d_state(int(g["state_seed"])))
    x = g["x"]
    masks = oinit.drop_masks_2d(int(g["mask_seed"]), x.shape[0], x.shape[2], x.shape[3])
    outs, dx, _ = _run(nets.dual_decoder_2d, sd, x, int(g["cot_seed"]), train=True, drop=masks)
    assert _relerr(outs[0], g["train_logits0"]) < 2e-5
    assert _relerr(outs[1], g["train_logits1"]) < 2e-5
    # fp32 train-mode grads through tiny-batch BN are ill-conditioned: the fp32 reference itself is
    # ~5e-3 from its fp64 run.  The fp32 oracle must be no further from fp64 truth than 3x that.
    ref_err = _relerr(g["train_dx"], g["train64_dx"])
    assert _relerr(dx, g["train64_dx"]) < max(3 * ref_err, 1e-4)
    # ... and in fp64 the oracle reproduces the reference algorithm exactly.
    sd64 = _sd(oinit.dual_decoder_2d_state(int(g["state_seed"])), dtype=torch.float64)
    o64, dx64, _ = _run(nets.dual_decoder_2d, sd64, x, int(g["cot_seed"]), dtype=torch.float64, train=True, drop=masks)
    assert _relerr(o64[0], g["train64_logits0"]) < 1e-6
    assert _relerr(dx64, g["train64_dx"]) < 1e-6
    for i, n in enumerate(g["grad_pick_names"]):
        assert _relerr(sd64[str(n)].grad, g["train64_grad_pick%d" % i]) < 1e-6, n
    for k in ("encoder.in_conv.conv_conv.1", "encoder.down3.maxpool_conv.1.conv_conv.5", "decoder2.up4.conv.conv_conv.1"):
        _close(sd[k + ".running_mean"], g["after_rm_" + k], 1e-5, 1e-6)
        _close(sd[k + ".running_var"], g["after_rv_" + k], 1e-5, 1e-6)
    assert int(sd["encoder.in_conv.conv_conv.1.num_batches_tracked"]) == 1


def test_dualdecoder2d_fullsize_eval(golden_dir):
    g = _load(golden_dir, "dualdecoder2d_256.npz")
    sd = _sd(oinit.dual_decoder_2d_state(int(g["state_seed"])), grad=False)
    x = torch.rand(1, 1, 256, 256, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    with torch.no_grad():
        o1, o2 = nets.dual_decoder_2d(sd, x, train=False)
    assert _relerr(o1[:, :, ::4, ::4], g["logits0_sub"]) < 1e-5
    assert _relerr(o2[:, :, ::4, ::4], g["logits1_sub"]) < 1e-5


def test_dualdecoder3d_fullsize_eval(golden_dir):
    """config-3 shape: the oracle's eval logits at 1 x 112 x 112 x 80 against the imported reference (vnet.py:225-238),
    sub-sampled values and whole-tensor checksums."""
    g = _load(golden_dir, "dualdecoder3d_112.npz")
    sd = _sd(oinit.dual_decoder_3d_state(int(g["state_seed"])), grad=False)
    x = torch.rand(1, 1, 112, 112, 80, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    with torch.no_grad():
        o1, o2 = nets.dual_decoder_3d(sd, x, train=False)
    assert _relerr(o1[:, :, ::4, ::4, ::4], g["logits0_sub"]) < 1e-5
    assert _relerr(o2[:, :, ::4, ::4, ::4], g["logits1_sub"]) < 1e-5
    sums = np.array([o1.double().sum().item(), o2.double().sum().item(), o1.double().abs().sum().item(), o2.double().abs().sum().item()])
    assert np.allclose(sums, g["sums"], rtol=1e-5)


def test_unet2d(golden_dir):
    g = _load(golden_dir, "unet2d_32.npz")
    state = oinit.unet_2d_state(int(g["state_seed"]))
    assert list(state.keys()) == list(g["keys"])
    outs, dx, _ = _run(nets.unet_2d, _sd(state), g["x"], int(g["cot_seed"]), train=False)
    assert _relerr(outs[0], g["eval_logits0"]) < 1e-5
    assert _relerr(dx, g["eval_dx"]) < 1e-4


def test_dualdecoder3d(golden_dir):
    g = _load(golden_dir, "dualdecoder3d_32.npz")
    state = oinit.dual_decoder_3d_state(int(g["state_seed"]))
    assert list(state.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in state.values()] == list(g["shapes"])
    assert len(state) == 298
    sd = _sd(state)
    outs, dx, _ = _run(nets.dual_decoder_3d, sd, g["x"], int(g["cot_seed"]), train=False)
    assert _relerr(outs[0], g["eval_logits0"]) < 1e-5
    assert _relerr(outs[1], g["eval_logits1"]) < 1e-5
    assert _relerr(dx, g["eval_dx"]) < 1e-4
    for i, n in enumerate(g["grad_pick_names"]):
        assert _relerr(sd[str(n)].grad, g["eval_grad_pick%d" % i]) < 1e-4, n
    # train mode with injected Dropout3d masks
    sd = _sd(state)
    masks = oinit.drop_masks_3d(int(g["mask_seed"]), g["x_train"].shape[0])
    outs, dx, _ = _run(nets.dual_decoder_3d, sd, g["x_train"], int(g["cot_seed"]), train=True, drop=masks)
    assert _relerr(outs[0], g["train_logits0"]) < 2e-5
    assert _relerr(outs[1], g["train_logits1"]) < 2e-5
    ref_err = _relerr(g["train_dx"], g["train64_dx"])
    assert _relerr(dx, g["train64_dx"]) < max(3 * ref_err, 1e-4)
    sd64 = _sd(state, dtype=torch.float64)
    o64, dx64, _ = _run(nets.dual_decoder_3d, sd64, g["x_train"], int(g["cot_seed"]), dtype=torch.float64, train=True, drop=masks)
    assert _relerr(o64[0], g["train64_logits0"]) < 1e-6
    assert _relerr(dx64, g["train64_dx"]) < 1e-6
    for i, n in enumerate(g["grad_pick_names"]):
        assert _relerr(sd64[str(n)].grad, g["train64_grad_pick%d" % i]) < 1e-6, n
    for k in ("encoder.block_one.conv.1", "decoder1.block_six_up.conv.2", "decoder2.block_eight_up.conv.1"):
        _close(sd[k + ".running_mean"], g["after_rm_" + k], 1e-5, 1e-6)
        _close(sd[k + ".running_var"], g["after_rv_" + k], 1e-5, 1e-6)


def test_vnet_and_unet3d(golden_dir):
    g = _load(golden_dir, "vnet_16.npz")
    state = oinit.vnet_state(int(g["state_seed"]))
    assert list(state.keys()) == list(g["keys"])
    outs, dx, _ = _run(nets.vnet_3d, _sd(state), g["x"], int(g["cot_seed"]), train=False, has_dropout=False)
    assert _relerr(outs[0], g["eval_logits0"]) < 1e-5
    assert _relerr(dx, g["eval_dx"]) < 1e-4
    g = _load(golden_dir, "unet3d_32.npz")
    state = oinit.unet_3d_state(int(g["state_seed"]))
    assert list(state.keys()) == list(g["keys"])
    assert len(state) == 38
    with torch.no_grad():
        o = nets.unet_3d(state, torch.from_numpy(g["x"]))
    assert _relerr(o, g["eval_logits0"]) < 1e-5


def test_dualdecoder2d_plus_and_same_variants(golden_dir):
    """decoder_type 'plus' (Decoder_plus, additive skips) and 'same' (unet.py:270-275) against the reference."""
    g = _load(golden_dir, "dualdecoder2d_variants_32.npz")
    x = g["x"]
    masks = oinit.drop_masks_2d(int(g["mask_seed"]), x.shape[0], x.shape[2], x.shape[3])
    for dt in ("plus", "same"):
        state = oinit.dual_decoder_2d_state(int(g[dt + "_state_seed"]), decoder_type=dt)
        sd = _sd(state)
        outs, dx, _ = _run(nets.dual_decoder_2d, sd, x, int(g["cot_seed"]), train=False)
        assert _relerr(outs[0], g[dt + "_eval_logits0"]) < 1e-5 and _relerr(outs[1], g[dt + "_eval_logits1"]) < 1e-5
        assert _relerr(dx, g[dt + "_eval_dx"]) < 1e-4
        sd64 = _sd(state, dtype=torch.float64)
        outs, dx, _ = _run(nets.dual_decoder_2d, sd64, x, int(g["cot_seed"]), dtype=torch.float64, train=True, drop=masks)
        assert _relerr(outs[1], g[dt + "_train64_logits1"]) < 1e-6
        assert _relerr(dx, g[dt + "_train64_dx"]) < 1e-5
        for i, n in enumerate(g["grad_pick_names"]):
            assert _relerr(sd64[str(n)].grad, g["%s_train64_grad_pick%d" % (dt, i)]) < 1e-5, (dt, n)


def test_oracle_mix_loss_and_generate_mask_against_the_reference_functions(golden_dir):
    """H13 / H14: tests/golden/train_plumbing.npz holds outputs of the reference's OWN mix_loss / generate_mask
    (train_ours_2D.py:198-216, 91-101; executed from the script's source in the build container by oracle/gen_golden.py
    with the build's DiceLoss_bcp injected -- a partial pin: the Dice term itself is the build's definition)."""
    import os
    import numpy as np
    import torch
    from oracle import train_step as ots
    z = np.load(os.path.join(golden_dir, "train_plumbing.npz"))
    logits, img_l, patch_l = torch.from_numpy(z["logits"]), torch.from_numpy(z["img_l"]), torch.from_numpy(z["patch_l"])
    y0, x0, bh, bw = (int(v) for v in z["box"])
    N, _, H, W = logits.shape
    assert (bh, bw) == (int(H * 2 / 3), int(W * 2 / 3))
    mask, loss_mask = ots.box_masks(N, H, W, y0, x0)
    assert np.array_equal(mask.long().numpy(), z["mask"]) and np.array_equal(loss_mask.long().numpy(), z["loss_mask"])
    for tag, kw in (("lab", dict(u_weight=0.5)), ("unlab", dict(u_weight=0.5, unlab=True)), ("w", dict(l_weight=0.7, u_weight=0.3))):
        lg = logits.clone().double().requires_grad_(True)
        li, lp, tot = ots.mix_loss(lg, img_l, patch_l, loss_mask.double(), **kw)
        tot.backward()
        assert np.allclose(np.array([float(li), float(lp), float(tot)]), z["%s_losses" % tag], rtol=1e-9, atol=1e-12)
        assert np.allclose(lg.grad.numpy(), z["%s_dlogits" % tag], rtol=1e-5, atol=1e-9)
