"""Native counterparts of the reference's models/ package (same module and class names)."""
from .phiseg import PHISeg  # noqa: F401
from .unet import Unet  # noqa: F401
from .probabilistic_unet import ProbabilisticUnet  # noqa: F401
