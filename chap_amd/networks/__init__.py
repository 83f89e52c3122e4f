from .net_factory import net_factory  # noqa: F401
from .net_factory_3d import net_factory_3d  # noqa: F401
from .unet import DualDecoder, UNet  # noqa: F401
from .vnet import DualDecoder3d, VNet  # noqa: F401
from .unet_3D import unet_3D  # noqa: F401
