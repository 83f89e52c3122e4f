"""`networks` shim: put `chap_amd/shim` on PYTHONPATH and the reference's scripts resolve their imports
(`from networks.net_factory import net_factory`, code/test_2D_fully.py:14, code/train_ours_2D.py:22;
`from networks.net_factory_3d import net_factory_3d`, code/test_LA.py:4; `from networks.unet_3D import unet_3D`,
code/test_3D.py:8; `from networks.unet import DualDecoder`, `from networks.vnet import VNet`) to the MI355X
implementations in chap_amd.networks -- the scripts themselves stay unchanged."""
