"""3D U-Net used by the reference's inference script (code/test_3D.py:8,27): same class name,
constructor signature and state-dict keys as code/networks/unet_3D.py:20-100 with
UnetConv3 / UnetUp3_CT from code/networks/utils.py:99-123,260-276:

    UnetConv3   (Conv3d 3^3 +b -> InstanceNorm3d (no affine) -> ReLU) x 2
    MaxPool3d(2) between stages, trilinear x2 (align_corners=False) + cat[skip, up] + UnetConv3 on the way up,
    final Conv3d 1^3; Dropout(p=0.3) after `center` and before `final` (inactive in eval mode).

Inference only (test_single_case runs it under torch.no_grad() with batch 1): InstanceNorm statistics are
the batch-of-one statistics of the fused conv epilogue; larger batches are run sample by sample.
"""
import torch
import torch.nn as nn

from ..engine import Op, Program
from .base import ChapNet, holder


def _kaiming_conv(cin, cout, k):
    # init_weights(m, 'kaiming') (networks_other.py:40-49): kaiming_normal_(a=0, mode='fan_in') on every Conv
    m = nn.Conv3d(cin, cout, k, 1, k // 2)
    nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
    return m


def _unet_conv3(cin, cout):
    # state-dict children: conv1.0 / conv2.0 (index 1 = InstanceNorm3d without parameters, 2 = ReLU)
    return holder(conv1=holder(_0=_kaiming_conv(cin, cout, 3)), conv2=holder(_0=_kaiming_conv(cout, cout, 3)))


class unet_3D(ChapNet):
    dims = 3

    def __init__(self, feature_scale=4, n_classes=21, is_deconv=True, in_channels=3, is_batchnorm=True):
        super().__init__()
        if in_channels != 1:
            raise NotImplementedError("chap_amd: in_channels=%d (single-channel volumes only)" % in_channels)
        self.in_channels, self.is_batchnorm, self.feature_scale = in_channels, is_batchnorm, feature_scale
        f = [int(x / feature_scale) for x in (64, 128, 256, 512, 1024)]
        if f[0] != 16:
            raise NotImplementedError("chap_amd: feature_scale=%s (first stage must have 16 filters)" % feature_scale)
        self.conv1 = _unet_conv3(in_channels, f[0])
        self.conv2 = _unet_conv3(f[0], f[1])
        self.conv3 = _unet_conv3(f[1], f[2])
        self.conv4 = _unet_conv3(f[2], f[3])
        self.center = _unet_conv3(f[3], f[4])
        for i in range(4, 0, -1):
            self.add_module("up_concat%d" % i, holder(conv=_unet_conv3(f[i] + f[i - 1], f[i - 1])))
        self.final = nn.Conv3d(f[0], n_classes, 1)
        nn.init.kaiming_normal_(self.final.weight.data, a=0, mode="fan_in")   # unet_3D.py:66-70 applies it to every Conv3d
        ops = []

        def block(pre, srcs, out, cin, cout, first=False):
            kw = dict(w=pre + ".conv1.0.weight", b=pre + ".conv1.0.bias", inorm=True, cin=cin, cout=cout)
            if first:
                ops.append(Op("c1", out + ".a", [], **kw))
            else:
                ops.append(Op("conv", out + ".a", srcs, ksize=3, **kw))
            ops.append(Op("conv", out, [out + ".a"], ksize=3, w=pre + ".conv2.0.weight", b=pre + ".conv2.0.bias", inorm=True, cin=cout, cout=cout))

        block("conv1", None, "c1", 1, f[0], first=True)
        for i in range(2, 5):
            ops.append(Op("pool", "p%d" % i, ["c%d" % (i - 1)]))
            block("conv%d" % i, ["p%d" % i], "c%d" % i, f[i - 2], f[i - 1])
        ops.append(Op("pool", "p5", ["c4"]))
        block("center", ["p5"], "u5", f[3], f[4])
        for i in range(4, 0, -1):
            ops.append(Op("up", "u%d.hi" % (i + 1), ["u%d" % (i + 1)], half_pixel=True))
            block("up_concat%d.conv" % i, ["c%d" % i, "u%d.hi" % (i + 1)], "u%d" % i, f[i] + f[i - 1], f[i - 1])
        ops.append(Op("conv", "logits", ["u1"], ksize=1, w="final.weight", b="final.bias", cin=f[0], cout=n_classes, head=True))
        self._finish_init(Program(3, ops, ["logits"]))

    def forward(self, inputs):
        if self.training:
            raise NotImplementedError("chap_amd: unet_3D is built for inference (eval mode) only")
        with torch.no_grad():
            outs = [self._run(inputs[i:i + 1])[0] for i in range(inputs.shape[0])]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    @staticmethod
    def apply_argmax_softmax(pred):
        return torch.softmax(pred, dim=1)
