"""Deterministic state-dict recipes for parity tests (TEST INFRASTRUCTURE ONLY).

Key names and shapes follow the reference checkpoint format (SURVEY.md section 8b;
`torch.save(model.state_dict())`, code/train_ours_2D.py:428-435).  gen_golden.py
loads these dicts into the *imported reference modules* with strict=True, which
pins the names/shapes; the values are a seed recipe so no weight file is shipped.
"""
import math
from collections import OrderedDict

import torch

FT_2D = (16, 32, 64, 128, 256)


class _Gen:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.sd = OrderedDict()

    def conv(self, key, shape, fan_in, bias_n):
        std = math.sqrt(2.0 / fan_in)
        self.sd[key + ".weight"] = torch.randn(shape, generator=self.g) * std
        self.sd[key + ".bias"] = torch.randn(bias_n, generator=self.g) * 0.1

    def bn(self, key, c):
        self.sd[key + ".weight"] = torch.rand(c, generator=self.g) + 0.5
        self.sd[key + ".bias"] = torch.randn(c, generator=self.g) * 0.1
        self.sd[key + ".running_mean"] = torch.randn(c, generator=self.g) * 0.1
        self.sd[key + ".running_var"] = torch.rand(c, generator=self.g) + 0.5
        self.sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _block2d(g, pre, cin, cout):
    g.conv(pre + ".0", (cout, cin, 3, 3), cin * 9, cout)
    g.bn(pre + ".1", cout)
    g.conv(pre + ".4", (cout, cout, 3, 3), cout * 9, cout)
    g.bn(pre + ".5", cout)


def _encoder2d(g, in_chns):
    _block2d(g, "encoder.in_conv.conv_conv", in_chns, FT_2D[0])
    for i in range(1, 5):
        _block2d(g, "encoder.down%d.maxpool_conv.1.conv_conv" % i, FT_2D[i - 1], FT_2D[i])


def _decoder2d(g, root, n_class, bilinear, plus=False):
    for k in range(1, 5):
        c1, c2 = FT_2D[5 - k], FT_2D[4 - k]
        up = "%s.up%d" % (root, k)
        if bilinear:
            g.conv(up + ".conv1x1", (c2, c1, 1, 1), c1, c2)
        else:
            g.conv(up + ".up", (c1, c2, 2, 2), c1, c2)
        _block2d(g, up + ".conv.conv_conv", c2 if plus else 2 * c2, c2)
    g.conv(root + ".out_conv", (n_class, FT_2D[0], 3, 3), FT_2D[0] * 9, n_class)


def dual_decoder_2d_state(seed, in_chns=1, n_class=4, decoder_type="mcnet"):
    """202 tensors; decoder1 bilinear, decoder2 transposed-conv ('mcnet'), bilinear ('same') or bilinear with
    additive skips ('plus')  -- unet.py:270-275."""
    g = _Gen(seed)
    _encoder2d(g, in_chns)
    _decoder2d(g, "decoder1", n_class, True)
    _decoder2d(g, "decoder2", n_class, decoder_type != "mcnet", decoder_type == "plus")
    return g.sd


def unet_2d_state(seed, in_chns=1, n_class=4):
    g = _Gen(seed)
    _encoder2d(g, in_chns)
    _decoder2d(g, "decoder", n_class, True)
    return g.sd


# --------------------------------------------------------------------------- 3D
_STAGES = (("one", 1), ("two", 2), ("three", 3), ("four", 3), ("five", 3))
_DEC = (("five_up", "six", 3), ("six_up", "seven", 3), ("seven_up", "eight", 2), ("eight_up", "nine", 1))


def _vblock(g, pre, n, cin, cout):
    for s in range(n):
        g.conv("%s.conv.%d" % (pre, 3 * s), (cout, cin if s == 0 else cout, 3, 3, 3), (cin if s == 0 else cout) * 27, cout)
        g.bn("%s.conv.%d" % (pre, 3 * s + 1), cout)


def _vencoder(g, in_chns, nf):
    c = in_chns
    for i, (name, n) in enumerate(_STAGES):
        co = nf * (2 ** i)
        _vblock(g, "encoder.block_" + name, n, c if i == 0 else co, co)
        if i < 4:
            dw = "encoder.block_%s_dw" % name
            g.conv(dw + ".conv.0", (2 * co, co, 2, 2, 2), co * 8, 2 * co)
            g.bn(dw + ".conv.1", 2 * co)
        c = co


def _vdecoder(g, root, n_class, nf, trilinear):
    for k, (upn, blk, n) in enumerate(_DEC):
        cin, cout = nf * (2 ** (4 - k)), nf * (2 ** (3 - k))
        up = "%s.block_%s" % (root, upn)
        if trilinear:
            g.conv(up + ".conv.1", (cout, cin, 3, 3, 3), cin * 27, cout)
            g.bn(up + ".conv.2", cout)
        else:
            g.conv(up + ".conv.0", (cin, cout, 2, 2, 2), cin, cout)
            g.bn(up + ".conv.1", cout)
        _vblock(g, "%s.block_%s" % (root, blk), n, cout, cout)
    g.conv(root + ".out_conv", (n_class, nf, 1, 1, 1), nf, n_class)


def dual_decoder_3d_state(seed, in_chns=1, n_class=2, nf=16):
    """298 tensors; decoder1 trilinear+conv, decoder2 transposed-conv."""
    g = _Gen(seed)
    _vencoder(g, in_chns, nf)
    _vdecoder(g, "decoder1", n_class, nf, True)
    _vdecoder(g, "decoder2", n_class, nf, False)
    return g.sd


def vnet_state(seed, in_chns=1, n_class=2, nf=16):
    g = _Gen(seed)
    _vencoder(g, in_chns, nf)
    _vdecoder(g, "decoder", n_class, nf, False)
    return g.sd


def unet_3d_state(seed, in_chns=1, n_class=2):
    """38 tensors (InstanceNorm3d holds none); filters 16-32-64-128-256 (unet_3D.py:29-30)."""
    g = _Gen(seed)
    f = (16, 32, 64, 128, 256)

    def uconv(pre, cin, cout):
        g.conv(pre + ".conv1.0", (cout, cin, 3, 3, 3), cin * 27, cout)
        g.conv(pre + ".conv2.0", (cout, cout, 3, 3, 3), cout * 27, cout)

    c = in_chns
    for i in range(4):
        uconv("conv%d" % (i + 1), c, f[i])
        c = f[i]
    uconv("center", f[3], f[4])
    for i in range(4, 0, -1):
        uconv("up_concat%d.conv" % i, f[i] + f[i - 1], f[i - 1])
    g.conv("final", (n_class, f[0], 1, 1, 1), f[0], n_class)
    return g.sd


def drop_masks_2d(seed, n, h, w):
    """Keep masks (uint8) for the five encoder Dropout sites of the 2D nets."""
    g = torch.Generator().manual_seed(seed)
    ps = (0.05, 0.1, 0.2, 0.3, 0.5)
    sites = ("encoder.in_conv.conv_conv",) + tuple("encoder.down%d.maxpool_conv.1.conv_conv" % i for i in range(1, 5))
    out = {}
    for i, (s, p) in enumerate(zip(sites, ps)):
        shape = (n, FT_2D[i], h >> i, w >> i)
        out[s] = (torch.rand(shape, generator=g) >= p).to(torch.uint8)
    return out


def drop_masks_3d(seed, n, nf=16):
    """Keep masks [N, C] (uint8) for the Dropout3d sites of the V-Net family."""
    g = torch.Generator().manual_seed(seed)
    return {"encoder.dropout": (torch.rand((n, nf * 16), generator=g) >= 0.5).to(torch.uint8),
            "decoder1.dropout": (torch.rand((n, nf), generator=g) >= 0.5).to(torch.uint8),
            "decoder2.dropout": (torch.rand((n, nf), generator=g) >= 0.5).to(torch.uint8),
            "decoder.dropout": (torch.rand((n, nf), generator=g) >= 0.5).to(torch.uint8)}
