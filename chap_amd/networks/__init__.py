from .net_factory import net_factory  # noqa: F401
from .unet import DualDecoder, UNet  # noqa: F401
