"""chap_amd: MI355X-native (gfx950) implementation of the CHAP training hot path.

Python host code mirroring the reference's nn.Module / net_factory interface
(`chap_amd.networks`) over a C-ABI shared library of hand-written HIP kernels
(`chap_amd/libchap_hip.so`, declared in include/chap_hip.h).  No CPU fallback.
"""
__version__ = "0.1.0"
