"""Native counterparts of the reference's models/ package (same module and class names)."""
from .phiseg import PHISeg  # noqa: F401
from .unet import Unet  # noqa: F401
from .probabilistic_unet import ProbabilisticUnet  # noqa: F401
from . import phiseg3D  # noqa: F401,E402
from .phiseg3D import PHISeg3D  # noqa: F401,E402
