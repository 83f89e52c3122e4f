"""3D V-Net family through the drop-in nn.Module boundary on the GPU against the golden fixtures
(outputs of the imported reference).  fp32 mode: logits 1e-4 rel (north_star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chap_amd.networks import DualDecoder3d, VNet, net_factory_3d
from oracle import init as oinit
from tests.test_net2d_gpu import cosine, relerr, run_case

DEV = "cuda"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def chan_masks(masks):
    """oracle keep masks [N, C] (uint8) -> Dropout3d multipliers keep/(1-p), p = 0.5"""
    return {k: (v.float() * 2.0).to(DEV) for k, v in masks.items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dualdecoder3d_eval(golden_dir, dtype):
    g = _load(golden_dir, "dualdecoder3d_32.npz")
    m = net_factory_3d("dualdecoder", 1, 2, "test", DEV)
    m.load_state_dict(oinit.dual_decoder_3d_state(int(g["state_seed"])), strict=True)
    m.set_compute_dtype(dtype).eval()
    outs, dx = run_case(m, g["x"], int(g["cot_seed"]))
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    assert relerr(outs[0], g["eval_logits0"]) < tol
    assert relerr(outs[1], g["eval_logits1"]) < tol
    grads = dict(m.named_parameters())
    if dtype == torch.float32:
        assert relerr(dx, g["eval_dx"]) < 1e-2 and cosine(dx, g["eval_dx"]) > 0.99999
        for i, n in enumerate(g["grad_pick_names"]):
            assert relerr(grads[str(n)].grad, g["eval_grad_pick%d" % i]) < 2e-3, n
    else:
        assert cosine(dx, g["eval_dx"]) > 0.9


def test_dualdecoder3d_fullsize_eval(golden_dir):
    """BASELINE config 3's volume: eval logits of the HIP DualDecoder3d at 1 x 112 x 112 x 80 in fp32 against the IMPORTED
    reference's (tests/golden/dualdecoder3d_112.npz, oracle/gen_golden.py:gen_3d_full; vnet.py:225-238): every 4th voxel per
    axis within north_star's 1e-4, plus the whole-tensor checksums (a wrong brick / tile anywhere in the volume moves them)."""
    g = _load(golden_dir, "dualdecoder3d_112.npz")
    m = net_factory_3d("dualdecoder", 1, 2, "test", DEV)
    m.load_state_dict(oinit.dual_decoder_3d_state(int(g["state_seed"])), strict=True)
    m.eval()
    x = torch.rand(1, 1, 112, 112, 80, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    with torch.no_grad():
        o1, o2 = m(x)
    assert relerr(o1[:, :, ::4, ::4, ::4], g["logits0_sub"]) < 1e-4
    assert relerr(o2[:, :, ::4, ::4, ::4], g["logits1_sub"]) < 1e-4
    sums = np.array([o1.double().sum().item(), o2.double().sum().item(), o1.double().abs().sum().item(), o2.double().abs().sum().item()])
    assert np.allclose(sums, g["sums"], rtol=1e-4), (sums, g["sums"])
    # bf16 (the benchmarked mode) on the same volume: same label map away from ties, logits within bf16's resolution of the network
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        b1, b2 = m(x)
    assert relerr(b1[:, :, ::4, ::4, ::4], g["logits0_sub"]) < 5e-2 and relerr(b2[:, :, ::4, ::4, ::4], g["logits1_sub"]) < 5e-2
    assert (b1.argmax(1) == o1.argmax(1)).float().mean() > 0.99


def test_dualdecoder3d_train_injected(golden_dir):
    g = _load(golden_dir, "dualdecoder3d_32.npz")
    m = net_factory_3d("dualdecoder", 1, 2, "train", DEV)
    m.load_state_dict(oinit.dual_decoder_3d_state(int(g["state_seed"])), strict=True)
    m.train()
    x = g["x_train"]
    masks = oinit.drop_masks_3d(int(g["mask_seed"]), x.shape[0])
    outs, dx = run_case(m, x, int(g["cot_seed"]), drop_masks=chan_masks(masks))
    assert relerr(outs[0], g["train_logits0"]) < 1e-4
    assert relerr(outs[1], g["train_logits1"]) < 1e-4
    ref_err = relerr(g["train_dx"], g["train64_dx"])
    assert relerr(dx, g["train64_dx"]) < max(4 * ref_err, 2e-2)
    grads = dict(m.named_parameters())
    for i, n in enumerate(g["grad_pick_names"]):
        ref_e = relerr(g["train_grad_pick%d" % i], g["train64_grad_pick%d" % i])
        assert relerr(grads[str(n)].grad, g["train64_grad_pick%d" % i]) < max(4 * ref_e, 2e-2), n
    sd = m.state_dict()
    for k in ("encoder.block_one.conv.1", "decoder1.block_six_up.conv.2", "decoder2.block_eight_up.conv.1"):
        assert relerr(sd[k + ".running_mean"], g["after_rm_" + k]) < 1e-4
        assert relerr(sd[k + ".running_var"], g["after_rv_" + k]) < 1e-4


def test_vnet_eval_and_random_dropout(golden_dir):
    g = _load(golden_dir, "vnet_16.npz")
    m = net_factory_3d("vnet", 1, 2, "test", DEV)
    m.load_state_dict(oinit.vnet_state(int(g["state_seed"])), strict=True)
    m.eval()
    outs, dx = run_case(m, g["x"], int(g["cot_seed"]))
    assert relerr(outs[0], g["eval_logits0"]) < 1e-4
    assert relerr(dx, g["eval_dx"]) < 1e-2
    # train mode with the in-kernel Dropout3d masks: finite, and frozen() gives dL/dx only
    t = net_factory_3d("dualdecoder", 1, 2, "train", DEV).train()
    x = torch.rand(2, 1, 32, 32, 16, device=DEV, requires_grad=True)
    with t.frozen():
        a, b = t(x, update_stats=False)
        (a.sum() + b.sum()).backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0


def test_unet3d_inference(golden_dir):
    """unet_3D (InstanceNorm3d, MaxPool3d, trilinear align_corners=False) against the reference's logits,
    and the sliding-window recipe of test_3D_util.test_single_case on a padded volume."""
    from chap_amd.networks import unet_3D
    g = _load(golden_dir, "unet3d_32.npz")
    u = net_factory_3d("unet_3D", 1, 2, "test", DEV)
    u.load_state_dict(oinit.unet_3d_state(int(g["state_seed"])), strict=True)
    u.eval()
    x = torch.from_numpy(g["x"]).to(DEV)
    o = u(x)
    assert relerr(o, g["eval_logits0"]) < 1e-4
    o2 = u(torch.cat([x, x.flip(2)]))                  # batch of 2 = two batch-of-one passes (InstanceNorm)
    assert relerr(o2[:1], g["eval_logits0"]) < 1e-4
    with pytest.raises(NotImplementedError):
        u.train()(x)


def test_pool3d_and_half_pixel_upsample():
    import torch.nn.functional as F
    from chap_amd import ops
    from tests.test_kernels_gpu import cl, uncl
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 4, 6, 8, generator=g)
    out = torch.empty(1, 2, 3, 4, 16, device=DEV)
    ops.act_pool2(ops.Lazy(cl(x, torch.float32)), out, None, dims=3)
    assert relerr(uncl(out), F.max_pool3d(x, 2)) < 1e-6
    up = torch.empty(1, 8, 12, 16, 16, device=DEV)
    ops.upsample2x(ops.Lazy(cl(x, torch.float32)), up, dims=3, half_pixel=True)
    assert relerr(uncl(up), F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)) < 1e-5
