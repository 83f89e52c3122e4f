"""Functional CPU restatement of the reference networks (TEST INFRASTRUCTURE ONLY).

Every function takes a *state dict* (name -> tensor, the reference's checkpoint
format) instead of owning modules, and takes dropout keep-masks
as explicit inputs so the HIP path and the oracle can be driven with identical
randomness.  Pinned against the imported reference by tests/golden/*.npz
(see oracle/gen_golden.py).

Reference definitions followed (paths relative to /root/reference):
  2D  ConvBlock      code/networks/unet.py:44-60
      DownBlock      code/networks/unet.py:63-75
      UpBlock        code/networks/unet.py:78-99
      Encoder        code/networks/unet.py:125-151
      Decoder        code/networks/unet.py:153-190
      UpBlock_plus / Decoder_plus   code/networks/unet.py:100-122, 192-243
      DualDecoder    code/networks/unet.py:245-292
      UNet           code/networks/unet.py:498-552
  3D  ConvBlock      code/networks/vnet.py:8-34
      Downsampling   code/networks/vnet.py:70-94
      Upsampling     code/networks/vnet.py:97-125
      Encoder        code/networks/vnet.py:127-168
      Decoder        code/networks/vnet.py:170-223
      DualDecoder3d  code/networks/vnet.py:225-238
      VNet           code/networks/vnet.py:303-315
      unet_3D        code/networks/unet_3D.py:20-100, code/networks/utils.py:99-123,260-276
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default, unet.py:52

DROPOUT_2D = (0.05, 0.1, 0.2, 0.3, 0.5)  # unet.py:250
ENC_BLOCKS_2D = ("encoder.in_conv.conv_conv",
                 "encoder.down1.maxpool_conv.1.conv_conv",
                 "encoder.down2.maxpool_conv.1.conv_conv",
                 "encoder.down3.maxpool_conv.1.conv_conv",
                 "encoder.down4.maxpool_conv.1.conv_conv")


class Ctx:
    """Per-forward options shared by all layers."""

    def __init__(self, train, drop=None, update_stats=True):
        self.train = bool(train)
        self.drop = drop or {}          # site name -> keep mask (bool / 0-1)
        self.update_stats = update_stats


def _bn(sd, key, x, ctx):
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    if ctx.train:
        if ctx.update_stats:
            y = F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], True, BN_MOMENTUM, BN_EPS)
            nbt = sd.get(key + ".num_batches_tracked")
            if nbt is not None:
                nbt += 1
            return y
        return F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], False, BN_MOMENTUM, BN_EPS)


def _drop(x, site, p, ctx):
    """Inverted dropout with an injected keep mask (identity when none given)."""
    if not ctx.train or p <= 0.0:
        return x
    keep = ctx.drop.get(site)
    if keep is None:
        return x
    keep = keep.to(x.dtype)
    while keep.dim() < x.dim():      # Dropout3d masks are [N, C]
        keep = keep.unsqueeze(-1)
    return x * keep * (1.0 / (1.0 - p))


# --------------------------------------------------------------------------- 2D
def conv_block_2d(sd, pre, x, p, ctx):
    x = F.conv2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], padding=1)
    x = F.leaky_relu(_bn(sd, pre + ".1", x, ctx), LEAKY_SLOPE)
    x = _drop(x, pre, p, ctx)
    x = F.conv2d(x, sd[pre + ".4.weight"], sd[pre + ".4.bias"], padding=1)
    return F.leaky_relu(_bn(sd, pre + ".5", x, ctx), LEAKY_SLOPE)


def encoder_2d(sd, x, ctx, root="encoder"):
    feats = []
    for i, pre in enumerate(ENC_BLOCKS_2D):
        pre = pre.replace("encoder", root, 1)
        if i > 0:
            x = F.max_pool2d(x, 2)
        x = conv_block_2d(sd, pre, x, DROPOUT_2D[i], ctx)
        feats.append(x)
    return feats


def decoder_2d(sd, root, feats, ctx):
    bilinear = (root + ".up1.conv1x1.weight") in sd
    x = feats[4]
    for k in range(1, 5):
        up = "%s.up%d" % (root, k)
        skip = feats[4 - k]
        if bilinear:
            x = F.conv2d(x, sd[up + ".conv1x1.weight"], sd[up + ".conv1x1.bias"])
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        else:
            x = F.conv_transpose2d(x, sd[up + ".up.weight"], sd[up + ".up.bias"], stride=2)
        if sd[up + ".conv.conv_conv.0.weight"].shape[1] == skip.shape[1]:      # UpBlock_plus (unet.py:100-122): x2 + x1
            x = conv_block_2d(sd, up + ".conv.conv_conv", skip + x, 0.0, ctx)
        else:
            x = conv_block_2d(sd, up + ".conv.conv_conv", torch.cat([skip, x], 1), 0.0, ctx)
    return F.conv2d(x, sd[root + ".out_conv.weight"], sd[root + ".out_conv.bias"], padding=1)


def dual_decoder_2d(sd, x, train=False, drop=None, update_stats=True, with_feat=False):
    ctx = Ctx(train, drop, update_stats)
    feats = encoder_2d(sd, x, ctx)
    o1 = decoder_2d(sd, "decoder1", feats, ctx)
    o2 = decoder_2d(sd, "decoder2", feats, ctx)
    return (o1, o2, feats) if with_feat else (o1, o2)


def unet_2d(sd, x, train=False, drop=None, update_stats=True):
    ctx = Ctx(train, drop, update_stats)
    return decoder_2d(sd, "decoder", encoder_2d(sd, x, ctx), ctx)


# --------------------------------------------------------------------------- 3D (V-Net family)
VNET_STAGES = (("one", 1), ("two", 2), ("three", 3), ("four", 3), ("five", 3))
VNET_DEC = (("five_up", "six", 3), ("six_up", "seven", 3), ("seven_up", "eight", 2), ("eight_up", "nine", 1))


def _vnet_block(sd, pre, x, n_stages, ctx):
    for s in range(n_stages):
        x = F.conv3d(x, sd["%s.conv.%d.weight" % (pre, 3 * s)], sd["%s.conv.%d.bias" % (pre, 3 * s)], padding=1)
        x = F.relu(_bn(sd, "%s.conv.%d" % (pre, 3 * s + 1), x, ctx))
    return x


def vnet_encoder(sd, x, ctx, has_dropout):
    feats = []
    for i, (name, n) in enumerate(VNET_STAGES):
        x = _vnet_block(sd, "encoder.block_" + name, x, n, ctx)
        if i < 4:
            feats.append(x)
            dw = "encoder.block_%s_dw" % name
            x = F.conv3d(x, sd[dw + ".conv.0.weight"], sd[dw + ".conv.0.bias"], stride=2)
            x = F.relu(_bn(sd, dw + ".conv.1", x, ctx))
    if has_dropout:
        x = _drop(x, "encoder.dropout", 0.5, ctx)
    feats.append(x)
    return feats


def vnet_decoder(sd, root, feats, ctx, has_dropout):
    trilinear = (root + ".block_five_up.conv.1.weight") in sd and sd[root + ".block_five_up.conv.1.weight"].dim() == 5
    x = feats[4]
    for k, (upn, blk, n) in enumerate(VNET_DEC):
        up = "%s.block_%s" % (root, upn)
        if trilinear:   # Upsample(trilinear, AC) -> Conv3d 3^3 -> BN -> ReLU   (vnet.py:105-106)
            x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)
            x = F.conv3d(x, sd[up + ".conv.1.weight"], sd[up + ".conv.1.bias"], padding=1)
            x = F.relu(_bn(sd, up + ".conv.2", x, ctx))
        else:           # ConvTranspose3d k2 s2 -> BN -> ReLU                  (vnet.py:103)
            x = F.conv_transpose3d(x, sd[up + ".conv.0.weight"], sd[up + ".conv.0.bias"], stride=2)
            x = F.relu(_bn(sd, up + ".conv.1", x, ctx))
        x = x + feats[3 - k]
        x = _vnet_block(sd, "%s.block_%s" % (root, blk), x, n, ctx)
    if has_dropout:
        x = _drop(x, root + ".dropout", 0.5, ctx)
    return F.conv3d(x, sd[root + ".out_conv.weight"], sd[root + ".out_conv.bias"])


def dual_decoder_3d(sd, x, train=False, drop=None, update_stats=True, has_dropout=True):
    ctx = Ctx(train, drop, update_stats)
    feats = vnet_encoder(sd, x, ctx, has_dropout)
    return (vnet_decoder(sd, "decoder1", feats, ctx, has_dropout),
            vnet_decoder(sd, "decoder2", feats, ctx, has_dropout))


def vnet_3d(sd, x, train=False, drop=None, update_stats=True, has_dropout=True):
    ctx = Ctx(train, drop, update_stats)
    return vnet_decoder(sd, "decoder", vnet_encoder(sd, x, ctx, has_dropout), ctx, has_dropout)


# --------------------------------------------------------------------------- 3D U-Net (test_3D.py)
def _unetconv3(sd, pre, x):
    for c in ("conv1", "conv2"):
        x = F.conv3d(x, sd["%s.%s.0.weight" % (pre, c)], sd["%s.%s.0.bias" % (pre, c)], padding=1)
        x = F.relu(F.instance_norm(x, eps=1e-5))   # InstanceNorm3d, no affine (utils.py:105)
    return x


def unet_3d(sd, x, train=False, drop=None):
    """unet_3D.forward (unet_3D.py:72-94). Dropout(p=.3) sites: 'dropout1', 'dropout2'."""
    ctx = Ctx(train, drop)
    skips = []
    for i in range(1, 5):
        x = _unetconv3(sd, "conv%d" % i, x)
        skips.append(x)
        x = F.max_pool3d(x, 2)
    x = _unetconv3(sd, "center", x)
    x = _drop(x, "dropout1", 0.3, ctx)
    for i in range(4, 0, -1):
        up = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)
        skip = skips[i - 1]
        off = up.shape[2] - skip.shape[2]
        skip = F.pad(skip, 2 * [off // 2, off // 2, 0])
        x = _unetconv3(sd, "up_concat%d.conv" % i, torch.cat([skip, up], 1))
    x = _drop(x, "dropout2", 0.3, ctx)
    return F.conv3d(x, sd["final.weight"], sd["final.bias"])
