"""3D V-Net family with the reference's module interface (code/networks/vnet.py): same class names,
constructor/forward signatures, attribute tree and state-dict keys; forward/backward are one
hand-scheduled HIP program (chap_amd.engine) on NDHWC tensors.

    ConvBlock              vnet.py:8-34     (Conv3d 3^3 p1 +b -> BatchNorm3d -> ReLU) x n_stages
    DownsamplingConvBlock  vnet.py:70-94    Conv3d k2 s2 -> BN -> ReLU
    Upsampling_function    vnet.py:97-125   mode 0: ConvTranspose3d k2 s2 | mode 1: trilinear x2 (align_corners) + Conv3d 3^3; -> BN -> ReLU
    Encoder vnet.py:127-168   Decoder vnet.py:170-223 (skip ADD, Dropout3d(.5) on x5 / x9)
    DualDecoder3d vnet.py:225-238   VNet vnet.py:303-315
Only normalization='batchnorm', has_residual=False (what net_factory_3d builds) is implemented.
"""
import torch.nn as nn

from ..engine import Op, Program
from .base import ChapNet, holder

STAGES = (("one", 1), ("two", 2), ("three", 3), ("four", 3), ("five", 3))
DEC = (("five_up", "six", 3), ("six_up", "seven", 3), ("seven_up", "eight", 2), ("eight_up", "nine", 1))


def _numbered(**layers):
    m = nn.Module()
    for k, v in layers.items():
        m.add_module(k.lstrip("_"), v)
    return holder(conv=m)


def _conv_block(n_stages, cin, cout):
    layers = {}
    for s in range(n_stages):
        layers["_%d" % (3 * s)] = nn.Conv3d(cin if s == 0 else cout, cout, 3, padding=1)
        layers["_%d" % (3 * s + 1)] = nn.BatchNorm3d(cout)
    return _numbered(**layers)


def _down(cin, cout):
    return _numbered(_0=nn.Conv3d(cin, cout, 2, padding=0, stride=2), _1=nn.BatchNorm3d(cout))


def _up(cin, cout, mode):
    if mode == 0:
        return _numbered(_0=nn.ConvTranspose3d(cin, cout, 2, padding=0, stride=2), _1=nn.BatchNorm3d(cout))
    return _numbered(_1=nn.Conv3d(cin, cout, 3, padding=1), _2=nn.BatchNorm3d(cout))


def _encoder(n_channels, nf):
    enc = nn.Module()
    c = n_channels
    for i, (name, n) in enumerate(STAGES):
        co = nf * (2 ** i)
        enc.add_module("block_" + name, _conv_block(n, c if i == 0 else co, co))
        if i < 4:
            enc.add_module("block_%s_dw" % name, _down(co, 2 * co))
        c = co
    enc.add_module("dropout", nn.Dropout3d(p=0.5, inplace=False))
    return enc


def _decoder(n_classes, nf, up_type):
    dec = nn.Module()
    for k, (upn, blk, n) in enumerate(DEC):
        cin, cout = nf * (2 ** (4 - k)), nf * (2 ** (3 - k))
        dec.add_module("block_" + upn, _up(cin, cout, up_type))
        dec.add_module("block_" + blk, _conv_block(n, cout, cout))
    dec.add_module("out_conv", nn.Conv3d(nf, n_classes, 1, padding=0))
    dec.add_module("dropout", nn.Dropout3d(p=0.5, inplace=False))
    return dec


def build_program(n_classes, nf, decoders, has_dropout, n_channels=1):
    """decoders: list of (root, up_type)."""
    ops = []

    def block(pre, srcs, out, n, cin, cout, first=False, combine=0, drop=None):
        cur = srcs
        for s in range(n):
            name = out if s == n - 1 else "%s.s%d" % (out, s)
            kw = dict(w="%s.conv.%d.weight" % (pre, 3 * s), b="%s.conv.%d.bias" % (pre, 3 * s), bn="%s.conv.%d" % (pre, 3 * s + 1),
                      slope=0.0, cin=cin if s == 0 else cout, cout=cout, drop=drop if s == n - 1 else None)
            if first and s == 0:
                ops.append(Op("c1", name, [], **kw))
            else:
                ops.append(Op("conv", name, cur, ksize=3, combine=combine if s == 0 else 0, **kw))
            cur = [name]

    c = n_channels
    x = None
    for i, (name, n) in enumerate(STAGES):
        co = nf * (2 ** i)
        drop = ("encoder.dropout", 0.5, "chan") if (i == 4 and has_dropout) else None
        block("encoder.block_" + name, [x] if x else None, "b%d" % (i + 1), n, c if i == 0 else co, co, first=(i == 0), drop=drop)
        if i < 4:
            dw = "encoder.block_%s_dw" % name
            ops.append(Op("down", "d%d" % (i + 1), ["b%d" % (i + 1)], w=dw + ".conv.0.weight", b=dw + ".conv.0.bias", bn=dw + ".conv.1",
                          slope=0.0, cin=co, cout=2 * co))
            x = "d%d" % (i + 1)
        c = co
    heads = []
    for bi, (root, up_type) in enumerate(decoders):
        first_dec_op = len(ops)
        x = "b5"
        for k, (upn, blk, n) in enumerate(DEC):
            cin, cout = nf * (2 ** (4 - k)), nf * (2 ** (3 - k))
            up = "%s.block_%s" % (root, upn)
            u = "%s.u%d" % (root, k)
            if up_type == 1:
                ops.append(Op("up", u + ".hi", [x]))
                ops.append(Op("conv", u, [u + ".hi"], ksize=3, w=up + ".conv.1.weight", b=up + ".conv.1.bias", bn=up + ".conv.2", slope=0.0, cin=cin, cout=cout))
            else:
                ops.append(Op("deconv", u, [x], w=up + ".conv.0.weight", b=up + ".conv.0.bias", bn=up + ".conv.1", slope=0.0, cin=cin, cout=cout))
            out = "%s.x%d" % (root, 6 + k)
            drop = (root + ".dropout", 0.5, "chan") if (k == 3 and has_dropout) else None
            block("%s.block_%s" % (root, blk), [u, "b%d" % (4 - k)], out, n, cout, cout, combine=1, drop=drop)
            x = out
        ops.append(Op("conv", root + ".logits", [x], ksize=1, w=root + ".out_conv.weight", b=root + ".out_conv.bias", cin=nf, cout=n_classes, head=True))
        heads.append(root + ".logits")
        for op in ops[first_dec_op:]:
            op.branch = bi + 1
    return Program(3, ops, heads)


def _check(normalization, has_residual, n_channels):
    if normalization != "batchnorm" or has_residual:
        raise NotImplementedError("chap_amd: only normalization='batchnorm', has_residual=False (net_factory_3d's configuration) is built")
    if not 1 <= n_channels <= 16:
        raise NotImplementedError("chap_amd: n_channels=%d (1..16: the first layer runs on the input zero-padded to 16 channels)" % n_channels)


class DualDecoder3d(ChapNet):
    """forward(input) -> (out_seg1, out_seg2)   -- vnet.py:234-238. decoder1: trilinear+conv, decoder2: transposed conv."""

    dims = 3

    def __init__(self, n_channels=3, n_classes=2, n_filters=16, normalization="none", has_dropout=False, has_residual=False, args=None):
        super().__init__()
        _check(normalization, has_residual, n_channels)
        self.in_chns = n_channels
        self.encoder = _encoder(n_channels, n_filters)
        self.decoder1 = _decoder(n_classes, n_filters, 1)
        self.decoder2 = _decoder(n_classes, n_filters, 0)
        self.encoder.has_dropout = self.decoder1.has_dropout = self.decoder2.has_dropout = has_dropout
        self._finish_init(build_program(n_classes, n_filters, [("decoder1", 1), ("decoder2", 0)], has_dropout, n_channels))

    def forward(self, input, drop_masks=None, update_stats=True, grad_buffer=None):
        out = self._run(input, drop_masks=drop_masks, update_stats=update_stats, grad_buffer=grad_buffer)
        return out[0], out[1]


class VNet(ChapNet):
    """forward(input) -> out_seg   -- vnet.py:303-315 (decoder with transposed convs)."""

    dims = 3

    def __init__(self, n_channels=3, n_classes=2, n_filters=16, normalization="none", has_dropout=False, has_residual=False):
        super().__init__()
        _check(normalization, has_residual, n_channels)
        self.in_chns = n_channels
        self.encoder = _encoder(n_channels, n_filters)
        self.decoder = _decoder(n_classes, n_filters, 0)
        self.encoder.has_dropout = self.decoder.has_dropout = has_dropout
        self._finish_init(build_program(n_classes, n_filters, [("decoder", 0)], has_dropout, n_channels))

    def forward(self, input, drop_masks=None, update_stats=True, grad_buffer=None):
        return self._run(input, drop_masks=drop_masks, update_stats=update_stats, grad_buffer=grad_buffer)[0]
