"""net_factory with the reference signature (code/networks/net_factory.py:11-24); only the
branches on the CHAP hot path are built ('unet', 'dualdecoder')."""
from .unet import DualDecoder, UNet


def net_factory(net_type="unet", in_chns=1, class_num=3, device="cuda:0", args=None):
    if net_type == "unet":
        net = UNet(in_chns=in_chns, class_num=class_num).to(device)
    elif net_type == "dualdecoder":
        net = DualDecoder(in_chns=in_chns, class_num=class_num, args=args).to(device)
    else:
        net = None          # the reference returns None for unknown names (net_factory.py:23)
    return net
