from chap_amd.networks.unet_3D import unet_3D  # noqa: F401
