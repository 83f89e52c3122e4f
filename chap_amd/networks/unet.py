"""2D networks with the reference's module interface (code/networks/unet.py): same class names,
constructor/forward signatures, attribute tree and state-dict keys -- the forward/backward are one
hand-scheduled HIP program (chap_amd.engine), not torch.nn calls.

    ConvBlock   unet.py:44-60     Conv3x3+b -> BN -> LeakyReLU -> Dropout(p) -> Conv3x3+b -> BN -> LeakyReLU
    DownBlock   unet.py:63-75     MaxPool2d(2) -> ConvBlock
    UpBlock     unet.py:78-99     [Conv1x1 -> bilinear x2 (align_corners)] | ConvTranspose2d k2 s2 ; cat ; ConvBlock
    Encoder     unet.py:125-151   Decoder unet.py:153-190   DualDecoder unet.py:245-292   UNet unet.py:498-552
"""
import random

import torch
import torch.nn as nn

from .. import ops as hip_ops
from ..engine import Op, Program
from ..ops import Lazy
from .base import ChapNet, holder

FT = (16, 32, 64, 128, 256)
ENC_FEATURES = ("e0", "e1", "e2", "e3", "e4")      # value names of the five encoder outputs in the Program
DROP = (0.05, 0.1, 0.2, 0.3, 0.5)
SLOPE = 0.01


def _conv_block(cin, cout):
    # children "0","1","4","5" = conv, bn, conv, bn (2, 3, 6 are activation / dropout: no state)
    return holder(conv_conv=holder(_0=nn.Conv2d(cin, cout, 3, padding=1), _1=nn.BatchNorm2d(cout),
                                   _4=nn.Conv2d(cout, cout, 3, padding=1), _5=nn.BatchNorm2d(cout)))


def _encoder(in_chns):
    enc = holder(in_conv=_conv_block(in_chns, FT[0]))
    for i in range(1, 5):
        enc.add_module("down%d" % i, holder(maxpool_conv=holder(_1=_conv_block(FT[i - 1], FT[i]))))
    return enc


def _decoder(n_class, bilinear, plus=False):
    """Decoder (unet.py:153-190: skip and up-sampled features concatenated) or, plus=True, Decoder_plus
    (unet.py:192-243: UpBlock_plus adds them, so its ConvBlock takes c2 channels)."""
    dec = nn.Module()
    for k in range(1, 5):
        c1, c2 = FT[5 - k], FT[4 - k]
        up = nn.Module()
        if bilinear:
            up.add_module("conv1x1", nn.Conv2d(c1, c2, 1))
        else:
            up.add_module("up", nn.ConvTranspose2d(c1, c2, 2, stride=2))
        up.add_module("conv", _conv_block(c2 if plus else 2 * c2, c2))
        dec.add_module("up%d" % k, up)
    dec.add_module("out_conv", nn.Conv2d(FT[0], n_class, 3, padding=1))
    return dec


def _block_ops(ops, pre, src, out, cin, cout, drop_p, first=False, combine=0):
    mid = out + ".a"
    kw = dict(w=pre + ".0.weight", b=pre + ".0.bias", bn=pre + ".1", slope=SLOPE, cin=cin, cout=cout,
              drop=(pre, drop_p, "elem") if drop_p > 0 else None, combine=combine)
    if first:
        ops.append(Op("c1", mid, [], **kw))
    else:
        ops.append(Op("conv", mid, src, ksize=3, **kw))
    ops.append(Op("conv", out, [mid], ksize=3, w=pre + ".4.weight", b=pre + ".4.bias", bn=pre + ".5", slope=SLOPE, cin=cout, cout=cout))


def build_program(n_class, decoders, enc_root="encoder", in_chns=1):
    """decoders: list of (root name, bilinear?[, plus?])."""
    ops = []
    _block_ops(ops, enc_root + ".in_conv.conv_conv", None, "e0", in_chns, FT[0], DROP[0], first=True)
    for i in range(1, 5):
        ops.append(Op("pool", "p%d" % i, ["e%d" % (i - 1)]))
        _block_ops(ops, "%s.down%d.maxpool_conv.1.conv_conv" % (enc_root, i), ["p%d" % i], "e%d" % i, FT[i - 1], FT[i], DROP[i])
    heads = []
    for bi, (root, bilinear, *rest) in enumerate(decoders):
        plus = bool(rest and rest[0])
        first_dec_op = len(ops)
        x = "e4"
        for k in range(1, 5):
            c1, c2 = FT[5 - k], FT[4 - k]
            up = "%s.up%d" % (root, k)
            u = "%s.u%d" % (root, k)
            if bilinear:
                ops.append(Op("conv", u + ".lo", [x], ksize=1, w=up + ".conv1x1.weight", b=up + ".conv1x1.bias", cin=c1, cout=c2))
                ops.append(Op("up", u, [u + ".lo"]))
            else:
                ops.append(Op("deconv", u, [x], w=up + ".up.weight", b=up + ".up.bias", cin=c1, cout=c2))
            d = "%s.d%d" % (root, k)
            if plus:    # UpBlock_plus (unet.py:117-122): x = x2 + x1, summed while the conv loads its two sources
                _block_ops(ops, up + ".conv.conv_conv", ["e%d" % (4 - k), u], d, c2, c2, 0.0, combine=1)
            else:
                _block_ops(ops, up + ".conv.conv_conv", ["e%d" % (4 - k), u], d, 2 * c2, c2, 0.0)
            x = d
        ops.append(Op("conv", root + ".logits", [x], ksize=3, w=root + ".out_conv.weight", b=root + ".out_conv.bias",
                      cin=FT[0], cout=n_class, head=True))
        heads.append(root + ".logits")
        for op in ops[first_dec_op:]:
            op.branch = bi + 1          # decoders are independent of each other: parallel streams
    return Program(2, ops, heads)


class DualDecoder(ChapNet):
    """forward(x, with_feat=False, dropout=False, dropout_level=None, scores=None, comp_dropout=False)
    -> (out1, out2[, features])   -- unet.py:277-292.  decoder1 = bilinear, decoder2 per decoder_type."""

    dims = 2

    def __init__(self, in_chns, class_num, args):
        super().__init__()
        if not 1 <= in_chns <= 16:
            raise NotImplementedError("chap_amd: in_chns=%d (1..16: the first layer runs on the input zero-padded to 16 channels)" % in_chns)
        self.in_chns = in_chns
        self.decoder_type = args["decoder_type"]
        if self.decoder_type not in ("mcnet", "same", "plus"):
            raise ValueError("chap_amd: decoder_type=%r (same | plus | mcnet, unet.py:270-275)" % self.decoder_type)
        self.encoder = _encoder(in_chns)
        self.decoder1 = _decoder(class_num, True)
        bil2, plus2 = self.decoder_type in ("same", "plus"), self.decoder_type == "plus"      # unet.py:270-275
        self.decoder2 = _decoder(class_num, bil2, plus2)
        self._finish_init(build_program(class_num, [("decoder1", True), ("decoder2", bil2, plus2)], in_chns=in_chns))

    def forward(self, x, with_feat=False, dropout=False, dropout_level=None, scores=None, comp_dropout=False,
                drop_masks=None, update_stats=True, grad_buffer=None, drop_uniforms=None, drop_branches=None):
        perturb = None
        if dropout:         # unet.py:280-284: both decoders on torch.cat((feat, perturbed unlabeled half)), B + U samples
            perturb = self._channel_perturbation(x.shape[0], dropout_level, scores, comp_dropout, drop_uniforms, drop_branches)
        if with_feat:       # unet.py:289-290: also the five encoder features (materialised NCHW fp32, detached)
            out = self._run(x, drop_masks=drop_masks, update_stats=update_stats, want=list(ENC_FEATURES), grad_buffer=grad_buffer,
                            perturb=perturb)
            return out[0], out[1], list(out[2:])
        out = self._run(x, drop_masks=drop_masks, update_stats=update_stats, grad_buffer=grad_buffer, perturb=perturb)
        return out[0], out[1]

    def _channel_perturbation(self, B, level, scores, comp, uniforms, branches):
        """FilterDropout.perform_dropout (FilterDropout.py:45-89) on the lazy encoder features: the decoders' batch is
        torch.cat((feat, feat[B//2:])) and the two channel masks become that batch's chan_mul rows (chap_channel_drop),
        so the masked copies are never materialised.  drop_uniforms[idx] = (u1, u2) fp32 [U, C] and drop_branches[idx]
        replace the device RNG / random.randint draws (tests)."""
        if B % 2:
            raise ValueError("chap_amd: dropout=True needs an even batch (labeled_bs = bs // 2 masks, FilterDropout.py:55-60)")
        level = () if level is None else tuple(level)
        U = B - B // 2

        def perturb(vals, vdims):
            ov1, ov2 = {}, {}
            for idx, name in enumerate(ENC_FEATURES):
                lz = vals[name]
                assert lz.keep is None and lz.chan_mul is None
                raw, Cc = lz.raw, lz.C
                dev = raw.device
                big = torch.empty((B + U,) + tuple(raw.shape[1:]), dtype=raw.dtype, device=dev)
                big[:B].copy_(raw)
                big[B:].copy_(raw[B // 2:])
                mul1 = mul2 = None
                if idx in level:
                    if uniforms is not None:
                        u1, u2 = (u.to(dev, torch.float32).contiguous() for u in uniforms[idx])
                    else:
                        u1, u2 = torch.empty(U, Cc, device=dev), torch.empty(U, Cc, device=dev)
                        hip_ops.rand_uniform(u1, self._rng.next_seed(), seed_dev=self._rng.seed_dev)
                        hip_ops.rand_uniform(u2, self._rng.next_seed(), seed_dev=self._rng.seed_dev)
                    mul1, mul2 = torch.empty(B + U, Cc, device=dev), torch.empty(B + U, Cc, device=dev)
                    if scores is None:
                        hip_ops.channel_drop(mul1, mul2, u1, u2, B, "comp_binomial" if comp else "dropout2d")
                    else:
                        d_, h_, w_ = vdims[name]
                        pooled = hip_ops.sample_channel_sum(Lazy(raw[B // 2:], lz.scale, lz.shift, lz.act, lz.slope))
                        branch = (branches[idx] if branches is not None else random.randint(0, 1)) if comp else 0
                        hip_ops.channel_drop(mul1, mul2, u1, u2, B, "scores", pool_partial=pooled, npix=d_ * h_ * w_,
                                         grad_sim=scores[idx].to(dev, torch.float32).contiguous(), comp=comp, branch=branch)
                ov1[name] = Lazy(big, lz.scale, lz.shift, lz.act, lz.slope, chan_mul=mul1)
                ov2[name] = Lazy(big, lz.scale, lz.shift, lz.act, lz.slope, chan_mul=mul2)
            return {1: ov1, 2: ov2}, B + U
        return perturb


class UNet(ChapNet):
    """forward(x, with_feats=False) -> logits   -- unet.py:498-521 (net_type='unet')."""

    dims = 2

    def __init__(self, in_chns, class_num):
        super().__init__()
        if not 1 <= in_chns <= 16:
            raise NotImplementedError("chap_amd: in_chns=%d (1..16)" % in_chns)
        self.in_chns = in_chns
        self.encoder = _encoder(in_chns)
        self.decoder = _decoder(class_num, True)
        self._finish_init(build_program(class_num, [("decoder", True)], in_chns=in_chns))

    def forward(self, x, with_feats=False, drop_masks=None, update_stats=True, grad_buffer=None):
        if with_feats:      # unet.py:513-520 -> Decoder.forward(feature, True) (:187-190): (logits, the last decoder feature [N, 16, H, W])
            out = self._run(x, drop_masks=drop_masks, update_stats=update_stats, want=["decoder.d4"], grad_buffer=grad_buffer)
            return out[0], out[1]      # the feature is materialised NCHW fp32 and detached (like DualDecoder's with_feat features)
        return self._run(x, drop_masks=drop_masks, update_stats=update_stats, grad_buffer=grad_buffer)[0]
