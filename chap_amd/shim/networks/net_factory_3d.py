from chap_amd.networks.net_factory_3d import net_factory_3d  # noqa: F401
