"""unet-zoo_amd: MI355X-native (gfx950) forward/backward hot path of gigantenbein/UNet-Zoo.

Host side: Python classes that mirror the reference's model API (models/unet.py, models/phiseg.py,
models/probabilistic_unet.py) and drive hand-written HIP kernels in ``libuz_hip.so`` through a
C ABI (``include/uz_api.h``) with ctypes.  PyTorch-ROCm tensors are used for storage only.
There is no CPU fallback: constructing a model without the HIP library or a GPU raises.
"""
__version__ = "0.1.0"
