"""2D networks through the drop-in nn.Module boundary on the GPU against the golden fixtures
(outputs of the imported reference) and the CPU oracle.  Tolerances: fp32 mode logits 1e-4 rel
(north_star), gradients judged against the fp64 truth relative to the fp32 reference's own error;
bf16 mode looser (stated below)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chap_amd.networks import DualDecoder, UNet, net_factory
from oracle import init as oinit
from oracle import nets as onets

DEV = "cuda"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def relerr(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    if np.abs(b).max() < 1e-9:
        return np.abs(a).max()
    return np.abs(a - b).max() / np.abs(b).max()


def cosine(a, b):
    a = a.detach().double().cpu().numpy().ravel() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64).ravel()
    b = b.detach().double().cpu().numpy().ravel() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def cl_masks(masks):
    return {k: v.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(DEV) for k, v in masks.items()}


def run_case(model, x, cot_seed, **kw):
    x = torch.from_numpy(x).to(DEV).requires_grad_(True)
    outs = model(x, **kw)
    outs = outs if isinstance(outs, tuple) else (outs,)
    g = torch.Generator().manual_seed(cot_seed)
    cots = [torch.randn(o.shape, generator=g).to(DEV) for o in outs]
    model.zero_grad()
    torch.autograd.backward(outs, cots)
    return outs, x.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dualdecoder_eval(golden_dir, dtype):
    g = _load(golden_dir, "dualdecoder2d_64.npz")
    m = net_factory("dualdecoder", 1, 4, DEV, {"decoder_type": "mcnet"})
    m.load_state_dict(oinit.dual_decoder_2d_state(int(g["state_seed"])), strict=True)
    m.set_compute_dtype(dtype).eval()
    outs, dx = run_case(m, g["x"], int(g["cot_seed"]))
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert relerr(outs[0], g["eval_logits0"]) < tol
    assert relerr(outs[1], g["eval_logits1"]) < tol
    grads = dict(m.named_parameters())
    if dtype == torch.float32:
        assert relerr(dx, g["eval_dx"]) < 2e-3
        for i, n in enumerate(g["grad_pick_names"]):
            assert relerr(grads[str(n)].grad, g["eval_grad_pick%d" % i]) < 2e-3, n
        names = [str(n) for n in g["param_names"]]
        got = np.array([float(grads[n].grad.double().abs().sum()) for n in names])
        np.testing.assert_allclose(got, g["eval_grad_checks"][:, 1], rtol=5e-3, atol=1e-4)
    else:
        # bf16 storage (eps 4e-3) through this random-weight net amplifies to ~0.2-0.3 relative L2 on
        # gradients (the fp32 path, same code with T=float, is at 2e-6): direction is what is checked.
        assert cosine(dx, g["eval_dx"]) > 0.9
        for i, n in enumerate(g["grad_pick_names"]):
            assert cosine(grads[str(n)].grad, g["eval_grad_pick%d" % i]) > 0.9, n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dualdecoder_train_injected(golden_dir, dtype):
    g = _load(golden_dir, "dualdecoder2d_64.npz")
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV)
    m.load_state_dict(oinit.dual_decoder_2d_state(int(g["state_seed"])), strict=True)
    m.set_compute_dtype(dtype).train()
    x = g["x"]
    masks = oinit.drop_masks_2d(int(g["mask_seed"]), x.shape[0], x.shape[2], x.shape[3])
    outs, dx = run_case(m, x, int(g["cot_seed"]), drop_masks=cl_masks(masks))
    tol = 1e-4 if dtype == torch.float32 else 1.5e-1
    assert relerr(outs[0], g["train_logits0"]) < tol
    assert relerr(outs[1], g["train_logits1"]) < tol
    grads = dict(m.named_parameters())
    if dtype == torch.float32:
        # gradients: distance to the fp64 truth, relative to the fp32 reference's own distance
        # (tiny-batch BN: the fp32 reference itself is ~5e-3 from its fp64 run)
        ref_err = relerr(g["train_dx"], g["train64_dx"])
        assert relerr(dx, g["train64_dx"]) < max(4 * ref_err, 2e-2)
        for i, n in enumerate(g["grad_pick_names"]):
            ref_e = relerr(g["train_grad_pick%d" % i], g["train64_grad_pick%d" % i])
            e = relerr(grads[str(n)].grad, g["train64_grad_pick%d" % i])
            if np.abs(g["train64_grad_pick%d" % i]).max() < 1e-9:      # conv bias before train-mode BN: exactly 0
                assert e < 1e-3, n
            else:
                # fp32 rounding of the BN statistics (fixed-order sums since round 2, still fp32) is amplified ~1e5x here
                assert e < max(4 * ref_e, 2e-2), (n, e, ref_e)
    else:
        # ill-conditioned case (32 samples per channel at the bottleneck BN): bf16 keeps the direction only
        assert cosine(dx, g["train64_dx"]) > 0.6
    sd = m.state_dict()
    rt = 1e-4 if dtype == torch.float32 else 5e-2
    for k in ("encoder.in_conv.conv_conv.1", "encoder.down3.maxpool_conv.1.conv_conv.5", "decoder2.up4.conv.conv_conv.1"):
        assert relerr(sd[k + ".running_mean"], g["after_rm_" + k]) < rt
        assert relerr(sd[k + ".running_var"], g["after_rv_" + k]) < rt
    assert int(sd["encoder.in_conv.conv_conv.1.num_batches_tracked"]) == 1


def test_dualdecoder_fullsize_and_unet(golden_dir):
    g = _load(golden_dir, "dualdecoder2d_256.npz")
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).eval()
    m.load_state_dict(oinit.dual_decoder_2d_state(int(g["state_seed"])), strict=True)
    x = torch.rand(1, 1, 256, 256, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    with torch.no_grad():
        o1, o2 = m(x)
    assert relerr(o1[:, :, ::4, ::4], g["logits0_sub"]) < 1e-4
    assert relerr(o2[:, :, ::4, ::4], g["logits1_sub"]) < 1e-4
    g = _load(golden_dir, "unet2d_32.npz")
    u = net_factory("unet", 1, 4, DEV)
    u.load_state_dict(oinit.unet_2d_state(int(g["state_seed"])), strict=True)
    u.eval()
    outs, dx = run_case(u, g["x"], int(g["cot_seed"]))
    assert relerr(outs[0], g["eval_logits0"]) < 1e-4
    assert relerr(dx, g["eval_dx"]) < 1e-2      # max-norm; a LeakyReLU sign flip at |z|~1e-7 moves single pixels
    assert cosine(dx, g["eval_dx"]) > 0.99999


def test_unet_with_feats(golden_dir):
    """UNet.forward(x, with_feats=True) -> (logits, last decoder feature), unet.py:513-520, against the imported reference."""
    g = _load(golden_dir, "unet2d_feats_32.npz")
    u = net_factory("unet", 1, 4, DEV)
    u.load_state_dict(oinit.unet_2d_state(int(g["state_seed"])), strict=True)
    u.eval()
    with torch.no_grad():
        o, f = u(torch.from_numpy(g["x"]).to(DEV), True)
    assert tuple(f.shape) == tuple(g["feat"].shape)
    assert relerr(o, g["logits"]) < 1e-4 and relerr(f, g["feat"]) < 1e-4


def test_train_mode_random_dropout_and_frozen():
    """in-kernel RNG dropout: runs, is reproducible for a fixed seed state, and the frozen()
    context yields dL/dx without touching parameter gradients."""
    torch.manual_seed(3)
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).train()
    x = torch.rand(2, 1, 64, 64, device=DEV, requires_grad=True)
    o1, o2 = m(x)
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
    with m.frozen():
        a, b = m(x, update_stats=False)
        (a.sum() + b.sum()).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    assert all(p.grad is None or p.grad.abs().sum() == 0 for p in m.parameters())


def test_cpu_module_fails_loudly():
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"})
    with pytest.raises(Exception, match="no CPU fallback"):
        m(torch.rand(1, 1, 32, 32))


def test_with_feat_returns_encoder_features(golden_dir):
    g = _load(golden_dir, "dualdecoder2d_64.npz")
    state = oinit.dual_decoder_2d_state(int(g["state_seed"]))
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(DEV).eval()
    m.load_state_dict(state, strict=True)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        o1, o2, feats = m(x.to(DEV), with_feat=True)
        r1, r2, rf = onets.dual_decoder_2d(state, x, train=False, with_feat=True)
    assert relerr(o1, r1) < 1e-4 and len(feats) == 5
    for a, b in zip(feats, rf):
        assert a.shape == b.shape and relerr(a, b) < 1e-4


@pytest.mark.parametrize("decoder_type", ["plus", "same"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dualdecoder_plus_and_same(golden_dir, decoder_type, dtype):
    """decoder_type 'plus' (UpBlock_plus: skip + up-sampled, summed while the conv loads its two sources) and 'same'
    (unet.py:270-275): eval logits and gradients, train-mode logits and gradients against the reference's fp64 run."""
    g = _load(golden_dir, "dualdecoder2d_variants_32.npz")
    dt = decoder_type
    m = DualDecoder(1, 4, {"decoder_type": dt}).to(DEV)
    m.load_state_dict(oinit.dual_decoder_2d_state(int(g[dt + "_state_seed"]), decoder_type=dt), strict=True)
    m.set_compute_dtype(dtype).eval()
    x = g["x"]
    outs, dx = run_case(m, x, int(g["cot_seed"]))
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert relerr(outs[0], g[dt + "_eval_logits0"]) < tol and relerr(outs[1], g[dt + "_eval_logits1"]) < tol
    if dtype == torch.float32:
        assert relerr(dx, g[dt + "_eval_dx"]) < 2e-3
        got = np.array([float(p.grad.double().abs().sum()) for _, p in m.named_parameters()])
        np.testing.assert_allclose(got, g[dt + "_eval_grad_checks"][:, 1], rtol=5e-3, atol=1e-4)
    else:
        assert cosine(dx, g[dt + "_eval_dx"]) > 0.9
    m.train()
    masks = oinit.drop_masks_2d(int(g["mask_seed"]), x.shape[0], x.shape[2], x.shape[3])
    outs, dx = run_case(m, x, int(g["cot_seed"]), drop_masks=cl_masks(masks))
    assert relerr(outs[0], g[dt + "_train64_logits0"]) < (1e-4 if dtype == torch.float32 else 1.5e-1)
    assert relerr(outs[1], g[dt + "_train64_logits1"]) < (1e-4 if dtype == torch.float32 else 1.5e-1)
    if dtype == torch.float32:
        grads = dict(m.named_parameters())
        # train-mode BN over 8 values per channel at the bottleneck: fp32 gradients (the reference's own too, see
        # test_dualdecoder_train_injected) sit ~1e-2 from the fp64 run; direction to 1e-3, size to 1e-1
        assert relerr(dx, g[dt + "_train64_dx"]) < 1e-1 and cosine(dx, g[dt + "_train64_dx"]) > 0.999
        for i, n in enumerate(g["grad_pick_names"]):
            want = g["%s_train64_grad_pick%d" % (dt, i)]
            assert relerr(grads[str(n)].grad, want) < 1e-1 and cosine(grads[str(n)].grad, want) > 0.999, n


@pytest.mark.parametrize("dims", [2, 3])
def test_multi_channel_input(dims):
    """in_chns > 1 (net_factory's second argument, net_factory.py:11 / net_factory_3d.py:7; the CHAP runs use 1): the first layer runs on the
    input zero-padded to 16 channels in both dtypes.  Eval logits, input gradient and the first layer's weight gradient against the oracle."""
    from chap_amd.networks import net_factory_3d
    from oracle import nets as onets
    g = torch.Generator().manual_seed(41)
    if dims == 2:
        cin, x = 3, torch.rand(2, 3, 32, 32, generator=g)
        state = oinit.dual_decoder_2d_state(77, in_chns=cin)
        m = net_factory("dualdecoder", cin, 4, DEV, {"decoder_type": "mcnet"})
        ofn, wkey = onets.dual_decoder_2d, "encoder.in_conv.conv_conv.0.weight"
    else:
        cin, x = 2, torch.rand(1, 2, 16, 32, 16, generator=g)
        state = oinit.dual_decoder_3d_state(78, in_chns=cin)
        m = net_factory_3d("dualdecoder", cin, 2, "test", DEV)
        ofn, wkey = onets.dual_decoder_3d, "encoder.block_one.conv.0.weight"
    m.load_state_dict(state, strict=True)
    m.eval()
    sd = {k: v.clone() for k, v in state.items()}
    sd[wkey].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    o = ofn(sd, xr, train=False)
    cot = [torch.randn(t.shape, generator=g) for t in o]
    sum((a * b).sum() for a, b in zip(o, cot)).backward()
    xd = x.to(DEV).requires_grad_(True)
    h = m(xd)
    torch.autograd.backward(h, [c.to(DEV) for c in cot])
    for a, b in zip(h, o):
        assert relerr(a, b.detach()) < 1e-4
    assert relerr(xd.grad, xr.grad) < 1e-2 and cosine(xd.grad, xr.grad) > 0.9999
    assert relerr(dict(m.named_parameters())[wkey].grad, sd[wkey].grad) < (2e-3 if dims == 2 else 5e-3)      # measured 3D: 2.4e-3 (a sum over 8 192 voxels through 40 fp32 layers)
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        hb = m(x.to(DEV))
    assert relerr(hb[0], o[0].detach()) < 6e-2
    with pytest.raises(ValueError):
        m(torch.rand(1, 1, *x.shape[2:], device=DEV))
