from chap_amd.networks.vnet import DualDecoder3d, VNet  # noqa: F401
