from chap_amd.networks.net_factory import net_factory  # noqa: F401
