#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the IMPORTED reference networks.

Runs only in the build container (needs /root/reference).  The reference is
imported read-only from where it lies (PYTHONDONTWRITEBYTECODE=1, nothing is
copied); five third-party packages that the hot path never calls are replaced
by empty stub modules.  What is committed is DATA only: seeded
inputs, outputs, gradients and state-dict key/shape lists.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/code"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import init as oinit  # noqa: E402
from oracle import filter_dropout as ofd  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("fvcore"); _stub("fvcore.nn"); _stub("fvcore.nn.weight_init")
    sys.modules["fvcore"].nn = sys.modules["fvcore.nn"]
    sys.modules["fvcore.nn"].weight_init = sys.modules["fvcore.nn.weight_init"]
    _stub("thop", clever_format=None, profile=None)
    _stub("torchsummary", summary=None)
    _stub("detectron2"); _stub("detectron2.config", configurable=lambda f: f)
    _stub("detectron2.utils"); _stub("detectron2.utils.registry", Registry=lambda n: None)
    _stub("timm"); _stub("timm.models")
    _stub("timm.models.layers", DropPath=None, trunc_normal_=None, trunc_normal_tf_=None, to_2tuple=None)
    sys.path.insert(0, REF)
    from networks.unet import DualDecoder, UNet
    from networks.vnet import DualDecoder3d, VNet
    from networks.unet_3D import unet_3D
    from networks.FilterDropout import perform_dropout
    return dict(DualDecoder=DualDecoder, UNet=UNet, DualDecoder3d=DualDecoder3d, VNet=VNet,
                unet_3D=unet_3D, perform_dropout=perform_dropout)


class Injected(nn.Module):
    """Stands in for nn.Dropout / nn.Dropout3d on a *live reference instance* so the
    reference runs with a known keep mask (same inverted-dropout scaling)."""

    def __init__(self, keep, p):
        super().__init__()
        self.keep, self.p = keep, p

    def forward(self, x):
        if not self.training:
            return x
        k = self.keep.to(x.dtype)
        while k.dim() < x.dim():
            k = k.unsqueeze(-1)
        return x * k / (1.0 - self.p)


def _np(t):
    a = t.detach().cpu().numpy()
    # fp64 truth runs are stored rounded to fp32 (6e-8 rel): half the fixture bytes
    return a.astype(np.float32) if a.dtype == np.float64 and a.size > 16 else a


def _checks(named):
    """per-tensor (sum, abs-sum) table, float64."""
    return np.array([[float(t.double().sum()), float(t.double().abs().sum())] for _, t in named], dtype=np.float64)


def run_case(model, x, cot_seed, train, dtype=torch.float32):
    """forward (+ backward against a fixed random cotangent) -> dict of arrays."""
    model.train(train)
    model.to(dtype)
    x = x.clone().to(dtype).requires_grad_(True)
    outs = model(x)
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    g = torch.Generator().manual_seed(cot_seed)
    loss = 0
    for o in outs:
        loss = loss + (o * torch.randn(o.shape, generator=g).to(dtype)).sum()
    model.zero_grad()
    loss.backward()
    res = {"logits%d" % i: _np(o) for i, o in enumerate(outs)}
    res["loss"] = np.float64(loss.item())
    res["dx"] = _np(x.grad)
    params = list(model.named_parameters())
    res["grad_checks"] = _checks([(n, p.grad) for n, p in params])
    return res, params


def gen_2d(ref):
    torch.manual_seed(0)
    N, H, W = 2, 64, 64
    sd = oinit.dual_decoder_2d_state(101)
    m = ref["DualDecoder"](1, 4, {"decoder_type": "mcnet"})
    m.load_state_dict(sd, strict=True)          # pins key names + shapes
    x = torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(7))
    out = {"x": _np(x), "state_seed": 101, "cot_seed": 11, "mask_seed": 21,
           "keys": np.array(list(m.state_dict().keys())),
           "shapes": np.array([str(tuple(v.shape)) for v in m.state_dict().values()]),
           "param_names": np.array([n for n, _ in m.named_parameters()])}
    r, params = run_case(m, x, 11, train=False)
    out.update({"eval_" + k: v for k, v in r.items()})
    pick = ["encoder.in_conv.conv_conv.0.weight", "encoder.in_conv.conv_conv.1.weight",
            "encoder.down2.maxpool_conv.1.conv_conv.4.weight", "decoder1.up1.conv1x1.weight",
            "decoder2.up3.up.weight", "decoder2.up3.up.bias", "decoder1.out_conv.weight",
            "decoder2.up4.conv.conv_conv.5.bias", "encoder.down4.maxpool_conv.1.conv_conv.0.bias"]
    out["grad_pick_names"] = np.array(pick)
    pd = dict(params)
    for i, n in enumerate(pick):
        out["eval_grad_pick%d" % i] = _np(pd[n].grad)
    # train mode, injected dropout masks, BN batch stats + running-stat update
    masks = oinit.drop_masks_2d(21, N, H, W)
    blocks = [m.encoder.in_conv] + [getattr(m.encoder, "down%d" % i).maxpool_conv[1] for i in range(1, 5)]
    for (site, keep), blk, p in zip(masks.items(), blocks, (0.05, 0.1, 0.2, 0.3, 0.5)):
        blk.conv_conv[3] = Injected(keep, p)
    r, params = run_case(m, x, 11, train=True)
    out.update({"train_" + k: v for k, v in r.items()})
    pd = dict(params)
    for i, n in enumerate(pick):
        out["train_grad_pick%d" % i] = _np(pd[n].grad)
    sd_after = m.state_dict()
    for k in ("encoder.in_conv.conv_conv.1", "encoder.down3.maxpool_conv.1.conv_conv.5", "decoder2.up4.conv.conv_conv.1"):
        out["after_rm_" + k] = _np(sd_after[k + ".running_mean"])
        out["after_rv_" + k] = _np(sd_after[k + ".running_var"])
    # the same train-mode case in float64: fp32 train-mode gradients through tiny-batch BN are
    # ill-conditioned (~1e-2 rel), so the algorithm is pinned in fp64 and fp32 paths are judged
    # by their distance to this fp64 truth.
    m64 = ref["DualDecoder"](1, 4, {"decoder_type": "mcnet"})
    m64.load_state_dict(oinit.dual_decoder_2d_state(101), strict=True)
    blocks = [m64.encoder.in_conv] + [getattr(m64.encoder, "down%d" % i).maxpool_conv[1] for i in range(1, 5)]
    for (site, keep), blk, p in zip(masks.items(), blocks, (0.05, 0.1, 0.2, 0.3, 0.5)):
        blk.conv_conv[3] = Injected(keep, p)
    r, params = run_case(m64, x, 11, train=True, dtype=torch.float64)
    out.update({"train64_" + k: v for k, v in r.items()})
    pd = dict(params)
    for i, n in enumerate(pick):
        out["train64_grad_pick%d" % i] = _np(pd[n].grad)
    np.savez_compressed(os.path.join(OUT, "dualdecoder2d_64.npz"), **out)

    # full-size (config-1 shape) eval logits, subsampled, N=1
    m2 = ref["DualDecoder"](1, 4, {"decoder_type": "mcnet"})
    m2.load_state_dict(oinit.dual_decoder_2d_state(101), strict=True)
    m2.eval()
    x = torch.rand(1, 1, 256, 256, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        o1, o2 = m2(x)
    np.savez_compressed(os.path.join(OUT, "dualdecoder2d_256.npz"), x_seed=8, state_seed=101,
                        logits0_sub=_np(o1[:, :, ::4, ::4]), logits1_sub=_np(o2[:, :, ::4, ::4]),
                        sums=np.array([o1.double().sum().item(), o2.double().sum().item(),
                                       o1.double().abs().sum().item(), o2.double().abs().sum().item()]))

    # plain UNet (net_type='unet')
    mu = ref["UNet"](1, 4)
    mu.load_state_dict(oinit.unet_2d_state(103), strict=True)
    x = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(9))
    r, _ = run_case(mu, x, 12, train=False)
    np.savez_compressed(os.path.join(OUT, "unet2d_32.npz"), x=_np(x), state_seed=103, cot_seed=12,
                        keys=np.array(list(mu.state_dict().keys())), **{"eval_" + k: v for k, v in r.items()})

    # default-init recipe: manual_seed(1337) then construct (code/train_ours_2D.py:487,551)
    torch.manual_seed(1337)
    md = ref["DualDecoder"](1, 4, {"decoder_type": "mcnet"})
    np.savez_compressed(os.path.join(OUT, "dualdecoder2d_init1337.npz"),
                        keys=np.array(list(md.state_dict().keys())),
                        checks=_checks(list(md.state_dict().items())))


def gen_unet_feats(ref):
    """UNet.forward(x, with_feats=True) (unet.py:513-520): (logits, last decoder feature) of the imported reference, eval mode."""
    mu = ref["UNet"](1, 4)
    mu.load_state_dict(oinit.unet_2d_state(103), strict=True)
    mu.eval()
    x = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        o, f = mu(x, True)
    np.savez_compressed(os.path.join(OUT, "unet2d_feats_32.npz"), x=_np(x), state_seed=103, logits=_np(o), feat=_np(f))


def gen_2d_variants(ref):
    """decoder_type 'plus' and 'same' (unet.py:270-275): eval and train-mode logits, input and picked weight gradients."""
    N, H, W = 2, 32, 32
    x = torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(17))
    out = {"x": _np(x), "cot_seed": 13, "mask_seed": 25}
    pick = ["decoder2.up3.conv1x1.weight", "decoder2.up4.conv.conv_conv.0.weight", "decoder2.up3.conv.conv_conv.0.weight",
            "decoder2.up4.conv.conv_conv.5.weight", "encoder.down1.maxpool_conv.1.conv_conv.4.weight", "encoder.in_conv.conv_conv.0.weight"]
    out["grad_pick_names"] = np.array(pick)
    for dt, seed in (("plus", 111), ("same", 112)):
        m = ref["DualDecoder"](1, 4, {"decoder_type": dt})
        m.load_state_dict(oinit.dual_decoder_2d_state(seed, decoder_type=dt), strict=True)      # pins names + shapes
        out[dt + "_state_seed"] = seed
        r, params = run_case(m, x, 13, train=False)
        out.update({"%s_eval_%s" % (dt, k): v for k, v in r.items()})
        masks = oinit.drop_masks_2d(25, N, H, W)
        blocks = [m.encoder.in_conv] + [getattr(m.encoder, "down%d" % i).maxpool_conv[1] for i in range(1, 5)]
        for (site, keep), blk, p in zip(masks.items(), blocks, (0.05, 0.1, 0.2, 0.3, 0.5)):
            blk.conv_conv[3] = Injected(keep, p)
        r, params = run_case(m, x, 13, train=True, dtype=torch.float64)
        out.update({"%s_train64_%s" % (dt, k): v for k, v in r.items()})
        pd = dict(params)
        for i, n in enumerate(pick):
            out["%s_train64_grad_pick%d" % (dt, i)] = _np(pd[n].grad)
    np.savez_compressed(os.path.join(OUT, "dualdecoder2d_variants_32.npz"), **out)


def gen_3d(ref):
    N, D, H, W = 1, 32, 32, 16
    sd = oinit.dual_decoder_3d_state(201)
    m = ref["DualDecoder3d"](n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
    m.load_state_dict(sd, strict=True)
    x = torch.rand(N, 1, D, H, W, generator=torch.Generator().manual_seed(17))
    out = {"x": _np(x), "state_seed": 201, "cot_seed": 31, "mask_seed": 41,
           "keys": np.array(list(m.state_dict().keys())),
           "shapes": np.array([str(tuple(v.shape)) for v in m.state_dict().values()]),
           "param_names": np.array([n for n, _ in m.named_parameters()])}
    r, params = run_case(m, x, 31, train=False)
    out.update({"eval_" + k: v for k, v in r.items()})
    pick = ["encoder.block_one.conv.0.weight", "encoder.block_two_dw.conv.0.weight",
            "decoder1.block_seven_up.conv.1.weight", "decoder2.block_seven_up.conv.0.weight",
            "decoder2.block_nine.conv.1.weight", "decoder1.out_conv.weight"]
    out["grad_pick_names"] = np.array(pick)
    pd = dict(params)
    for i, n in enumerate(pick):
        out["eval_grad_pick%d" % i] = _np(pd[n].grad)
    N2 = 2
    x2 = torch.rand(N2, 1, D, H, W, generator=torch.Generator().manual_seed(18))
    masks = oinit.drop_masks_3d(41, N2)
    m.encoder.dropout = Injected(masks["encoder.dropout"], 0.5)
    m.decoder1.dropout = Injected(masks["decoder1.dropout"], 0.5)
    m.decoder2.dropout = Injected(masks["decoder2.dropout"], 0.5)
    r, params = run_case(m, x2, 31, train=True)
    out["x_train"] = _np(x2)
    out.update({"train_" + k: v for k, v in r.items()})
    pd = dict(params)
    for i, n in enumerate(pick):
        out["train_grad_pick%d" % i] = _np(pd[n].grad)
    sd_after = m.state_dict()
    for k in ("encoder.block_one.conv.1", "decoder1.block_six_up.conv.2", "decoder2.block_eight_up.conv.1"):
        out["after_rm_" + k] = _np(sd_after[k + ".running_mean"])
        out["after_rv_" + k] = _np(sd_after[k + ".running_var"])
    m64 = ref["DualDecoder3d"](n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
    m64.load_state_dict(oinit.dual_decoder_3d_state(201), strict=True)
    m64.encoder.dropout = Injected(masks["encoder.dropout"], 0.5)
    m64.decoder1.dropout = Injected(masks["decoder1.dropout"], 0.5)
    m64.decoder2.dropout = Injected(masks["decoder2.dropout"], 0.5)
    r, params = run_case(m64, x2, 31, train=True, dtype=torch.float64)
    out.update({"train64_" + k: v for k, v in r.items()})
    pd = dict(params)
    for i, n in enumerate(pick):
        out["train64_grad_pick%d" % i] = _np(pd[n].grad)
    np.savez_compressed(os.path.join(OUT, "dualdecoder3d_32.npz"), **out)

    mv = ref["VNet"](n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=False)
    mv.load_state_dict(oinit.vnet_state(203), strict=True)
    x = torch.rand(1, 1, 16, 16, 16, generator=torch.Generator().manual_seed(19))
    r, _ = run_case(mv, x, 32, train=False)
    np.savez_compressed(os.path.join(OUT, "vnet_16.npz"), x=_np(x), state_seed=203, cot_seed=32,
                        keys=np.array(list(mv.state_dict().keys())), **{"eval_" + k: v for k, v in r.items()})

    mu = ref["unet_3D"](n_classes=2, in_channels=1)
    mu.load_state_dict(oinit.unet_3d_state(205), strict=True)
    mu.eval()
    x = torch.rand(1, 1, 32, 32, 32, generator=torch.Generator().manual_seed(20))
    with torch.no_grad():
        o = mu(x)
    np.savez_compressed(os.path.join(OUT, "unet3d_32.npz"), x=_np(x), state_seed=205,
                        keys=np.array(list(mu.state_dict().keys())), eval_logits0=_np(o))

    torch.manual_seed(1337)
    md = ref["DualDecoder3d"](n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
    np.savez_compressed(os.path.join(OUT, "dualdecoder3d_init1337.npz"),
                        keys=np.array(list(md.state_dict().keys())),
                        checks=_checks(list(md.state_dict().items())))


def gen_3d_full(ref):
    """Full-size (config-3 shape) eval logits of the imported DualDecoder3d, N = 1 at 112 x 112 x 80 (vnet.py:225-238; the LA patch
    of test_LA.py:24): every 4th voxel per axis plus whole-tensor checksums."""
    m = ref["DualDecoder3d"](n_channels=1, n_classes=2, normalization="batchnorm", has_dropout=True)
    m.load_state_dict(oinit.dual_decoder_3d_state(201), strict=True)
    m.eval()
    x = torch.rand(1, 1, 112, 112, 80, generator=torch.Generator().manual_seed(28))
    with torch.no_grad():
        o1, o2 = m(x)
    np.savez_compressed(os.path.join(OUT, "dualdecoder3d_112.npz"), x_seed=28, state_seed=201,
                        logits0_sub=_np(o1[:, :, ::4, ::4, ::4]), logits1_sub=_np(o2[:, :, ::4, ::4, ::4]),
                        sums=np.array([o1.double().sum().item(), o2.double().sum().item(),
                                       o1.double().abs().sum().item(), o2.double().abs().sum().item()]))
    print("dualdecoder3d_112.npz: logits", tuple(o1.shape), "abs max", float(o1.abs().max()), float(o2.abs().max()))


class ScriptedDraws:
    """While active, every random draw FilterDropout.py makes comes from a scripted list of uniform tensors:
    torch.bernoulli(q) -> (u < q), Binomial(0.5).sample(shape) -> (u < 0.5), nn.Dropout2d(0.5)(x) -> x * 2 * (u < 0.5),
    random.randint(0, 1) -> the scripted branch.  Patches torch / random attributes only (restored on exit); the
    reference module itself is untouched."""

    def __init__(self, uniforms, branches):
        self.q = [u for pair in uniforms for u in pair]      # consumption order: level by level, mask 1 then mask 2
        self.branches = list(branches)
        self.pos = 0

    def _next(self, shape):
        u = self.q[self.pos]
        self.pos += 1
        assert tuple(u.shape) == tuple(shape), (u.shape, shape)
        return u

    def skip(self):
        self.pos += 1

    def __enter__(self):
        import random
        import torch.distributions.binomial as tdb
        draws = self

        class Drop2d(nn.Module):
            def __init__(self, p):
                super().__init__()
                assert p == 0.5

            def forward(self, x):
                return x * ((draws._next(x.shape[:2]) < 0.5).to(x.dtype) * 2.0)[..., None, None]

        self.saved = (torch.bernoulli, tdb.Binomial.sample, nn.Dropout2d, random.randint)
        torch.bernoulli = lambda q: (draws._next(q.shape) < q).to(q.dtype)

        def sample(self_, shape=torch.Size()):
            m = (draws._next(tuple(shape)) < 0.5).float()
            draws.skip()                                      # the complementary mask uses no second draw
            return m
        tdb.Binomial.sample = sample
        nn.Dropout2d = Drop2d
        random.randint = lambda a, b: draws.branches.pop(0)
        return self

    def __exit__(self, *a):
        import random
        import torch.distributions.binomial as tdb
        torch.bernoulli, tdb.Binomial.sample, nn.Dropout2d, random.randint = self.saved


FD_CASES = (  # name, scores given, comp, branch
    ("drop2d", False, False, 0), ("binom", False, True, 0), ("scores", True, False, 0),
    ("scores_comp0", True, True, 0), ("scores_comp1", True, True, 1))


def gen_filter_dropout(ref):
    """perform_dropout / scores_dropoutV2 / drop_based_on_prob of the imported reference under scripted draws, and
    DualDecoder.forward(dropout=True) on top of it."""
    feats, scores, uniforms = ofd.fd_inputs()
    level = [0, 1, 2, 4]                                            # level 3 stays unperturbed
    out = {"level": np.array(level)}
    B = feats[0].shape[0]
    for name, with_scores, comp, branch in FD_CASES:
        with ScriptedDraws([uniforms[i] for i in level], [branch] * 5):
            f1, f2 = ref["perform_dropout"]([f.clone() for f in feats], level, scores if with_scores else None, comp)
        for idx, (a, b, f) in enumerate(zip(f1, f2, feats)):
            unlab = f[B // 2:]
            for tag, t in (("m1", a), ("m2", b)):
                assert torch.equal(t[:B], f)
                m = t[B:, :, 0, 0] / unlab[:, :, 0, 0]
                assert torch.allclose(t[B:], m[..., None, None] * unlab, rtol=1e-6, atol=0)
                out["%s_L%d_%s" % (name, idx, tag)] = _np(m)
    # the whole dropout=True forward (train mode, injected encoder dropout), N = 4 -> 6 output samples
    torch.manual_seed(0)
    N, H, W = 4, 32, 32
    m = ref["DualDecoder"](1, 4, {"decoder_type": "mcnet"})
    m.load_state_dict(oinit.dual_decoder_2d_state(101), strict=True)
    masks = oinit.drop_masks_2d(23, N, H, W)
    blocks = [m.encoder.in_conv] + [getattr(m.encoder, "down%d" % i).maxpool_conv[1] for i in range(1, 5)]
    for (site, keep), blk, p in zip(masks.items(), blocks, (0.05, 0.1, 0.2, 0.3, 0.5)):
        blk.conv_conv[3] = Injected(keep, p)
    x = torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(9))
    m.train()
    lv = [0, 1, 2, 3, 4]
    with torch.no_grad(), ScriptedDraws(uniforms, [1] * 5):
        o1, o2 = m(x, False, True, lv, scores, True)
    out.update(fwd_x=_np(x), fwd_state_seed=101, fwd_mask_seed=23, fwd_logits1=_np(o1), fwd_logits2=_np(o2))
    # backward through the perturbed pass: eval mode (running statistics: well conditioned, fp32) and train mode in fp64
    pick = ["decoder1.up3.conv1x1.weight", "decoder2.up3.up.weight", "decoder2.up4.conv.conv_conv.0.weight",
            "decoder1.up3.conv.conv_conv.1.weight", "encoder.down4.maxpool_conv.1.conv_conv.5.bias",
            "encoder.down1.maxpool_conv.1.conv_conv.4.weight", "encoder.in_conv.conv_conv.0.weight"]
    out["bwd_pick_names"] = np.array(pick)
    out["bwd_cot_seed"] = 31
    for tag, train, dtype in (("eval", False, torch.float32), ("train64", True, torch.float64)):
        mm = ref["DualDecoder"](1, 4, {"decoder_type": "mcnet"})
        mm.load_state_dict(oinit.dual_decoder_2d_state(101), strict=True)
        blocks = [mm.encoder.in_conv] + [getattr(mm.encoder, "down%d" % i).maxpool_conv[1] for i in range(1, 5)]
        for (site, keep), blk, p in zip(masks.items(), blocks, (0.05, 0.1, 0.2, 0.3, 0.5)):
            blk.conv_conv[3] = Injected(keep, p)
        mm.train(train)
        mm.to(dtype)
        xx = x.clone().to(dtype).requires_grad_(True)
        with ScriptedDraws(uniforms, [1] * 5):
            o1, o2 = mm(xx, False, True, lv, scores, True)
        g = torch.Generator().manual_seed(31)
        loss = sum((o * torch.randn(o.shape, generator=g).to(dtype)).sum() for o in (o1, o2))
        loss.backward()
        pd = dict(mm.named_parameters())
        out["bwd_%s_logits1" % tag] = _np(o1)
        out["bwd_%s_dx" % tag] = _np(xx.grad)
        out["bwd_%s_grad_checks" % tag] = _checks([(n, p.grad) for n, p in mm.named_parameters()])
        for i, n in enumerate(pick):
            out["bwd_%s_grad_pick%d" % (tag, i)] = _np(pd[n].grad)
    np.savez_compressed(os.path.join(OUT, "filter_dropout.npz"), **out)


def _reference_functions(names, namespace):
    """The reference's training script cannot be imported (absent in-repo modules).  Its pure-torch helper
    functions can still be RUN: parse the script as text, take the FunctionDef nodes asked for and exec only those in a
    namespace we provide.  Build container only; nothing of the source is stored -- the outputs are."""
    import ast
    tree = ast.parse(open(os.path.join(REF, "train_ours_2D.py")).read())
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names)
    exec(compile(ast.Module(body=picked, type_ignores=[]), "train_ours_2D.py", "exec"), namespace)
    return [namespace[n] for n in names]


def gen_train_plumbing():
    """H13 / H14 pins: the reference's own `mix_loss` (train_ours_2D.py:198-216) and `generate_mask` (:91-101) executed on
    seeded inputs.  `losses.DiceLoss_bcp` is absent upstream, so the build's definition (oracle.train_step.dice_loss_bcp) is
    injected as `dice_loss`: the pin covers the weighting / masking / cross-entropy plumbing around it (a PARTIAL pin, said so
    in DESIGN.md).  generate_mask draws its box from numpy's RNG: seeded here, the offsets are stored with the masks."""
    import torch.nn.functional as F
    from oracle import train_step as ots

    def dice_loss(soft, target, mask):                       # losses.DiceLoss_bcp(n_classes=4)(soft, target[N,1,H,W], mask[N,1,H,W])
        return ots.dice_loss_bcp(soft, target.squeeze(1), mask.squeeze(1), soft.shape[1])

    ns = dict(torch=torch, nn=nn, F=F, np=np, dice_loss=dice_loss)
    mix_loss, generate_mask = _reference_functions(["mix_loss", "generate_mask"], ns)
    g = torch.Generator().manual_seed(2024)
    N, C, H, W = 3, 4, 24, 36
    out = {}
    logits = torch.randn(N, C, H, W, generator=g) * 2
    img_l = torch.randint(0, C, (N, H, W), generator=g)
    patch_l = torch.randint(0, C, (N, H, W), generator=g)
    np.random.seed(7)
    mask, loss_mask = generate_mask(torch.zeros(N, 1, H, W))
    ys, xs = np.where(mask.numpy() == 0)
    out.update(logits=_np(logits), img_l=img_l.numpy(), patch_l=patch_l.numpy(), mask=mask.numpy(), loss_mask=loss_mask.numpy(),
               box=np.array([ys.min(), xs.min(), ys.max() - ys.min() + 1, xs.max() - xs.min() + 1]))
    for tag, kw in (("lab", dict(u_weight=0.5)), ("unlab", dict(u_weight=0.5, unlab=True)), ("w", dict(l_weight=0.7, u_weight=0.3))):
        lg = logits.clone().double().requires_grad_(True)
        li, lp, tot = mix_loss(lg, img_l, patch_l, loss_mask.double(), **kw)
        tot.backward()
        out["%s_losses" % tag] = np.array([float(li), float(lp), float(tot)])
        out["%s_dlogits" % tag] = _np(lg.grad)
    np.savez_compressed(os.path.join(OUT, "train_plumbing.npz"), **out)
    print("train_plumbing.npz: mix_loss x3 weightings, generate_mask box", out["box"])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = import_reference()
    only = sys.argv[1:]                     # e.g. `gen_golden.py gen_3d_full`: (re)generate one fixture
    steps = [("gen_2d", lambda: gen_2d(ref)), ("gen_2d_variants", lambda: gen_2d_variants(ref)), ("gen_3d", lambda: gen_3d(ref)),
             ("gen_3d_full", lambda: gen_3d_full(ref)), ("gen_unet_feats", lambda: gen_unet_feats(ref)), ("gen_filter_dropout", lambda: gen_filter_dropout(ref)),
             ("gen_train_plumbing", gen_train_plumbing)]
    assert all(o in dict(steps) for o in only), only
    for name, fn in steps:
        if not only or name in only:
            fn()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
