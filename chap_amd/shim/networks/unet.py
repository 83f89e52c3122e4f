from chap_amd.networks.unet import DualDecoder, UNet  # noqa: F401
