"""unet-zoo_amd: MI355X-native (gfx950) forward/backward hot path of gigantenbein/UNet-Zoo.

Host side: Python classes that mirror the reference's model API (models/unet.py, models/phiseg.py,
models/probabilistic_unet.py) and drive hand-written HIP kernels in ``libuz_hip.so`` through a
C ABI (``include/uz_api.h``) with ctypes.  PyTorch-ROCm tensors are used for storage only.
There is no CPU fallback: constructing a model without the HIP library or a GPU raises.
"""
__version__ = "0.1.0"

import os as _os

# hipGraph replay of a tape with parallel branches (dependency lanes, bucket events) spreads the branches over the process's
# hardware queues; every cross-branch edge then is a cross-queue barrier.  Measured on MI355X / ROCm 7.2 (PHiSeg step, batch 32):
# GPU_MAX_HW_QUEUES = 2: 21.99 ms, 3: 22.02, 4 (the default): 22.42, 8: 35.2; with the overlapped data-parallel exchange 4
# queues give anything from 21.8 to 39 ms depending on stream creation order, 2 queues a stable 22.7.  The variable is read
# when the HIP runtime initialises (first device call), so setting it at import time is early enough.
def _hip_already_initialised():
    """True when this process has made a device call before importing the package: the two queue variables below are read
    once, when the HIP runtime initialises, so setting them now would silently do nothing."""
    try:
        import sys as _sys
        _t = _sys.modules.get("torch")
        return bool(_t is not None and _t.cuda.is_initialized())
    except Exception:
        return False


if _hip_already_initialised() and ("GPU_MAX_HW_QUEUES" not in _os.environ or "DEBUG_HIP_FORCE_GRAPH_QUEUES" not in _os.environ):
    import warnings as _warnings
    _warnings.warn("unet_zoo_amd was imported after the HIP runtime had initialised: the tuned queue configuration "
                   "(GPU_MAX_HW_QUEUES=3, DEBUG_HIP_FORCE_GRAPH_QUEUES=3) is NOT in force for this process. Import the package "
                   "(or export the two variables) before the first torch.cuda call; measured cost of the runtime defaults on "
                   "MI355X: +0.5 ms per PHiSeg step, +14 ms with the overlapped data-parallel exchange.", RuntimeWarning, stacklevel=2)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
# The runtime maps a graph's branches onto DEBUG_HIP_FORCE_GRAPH_QUEUES internal streams (default 4) but creates no more of
# them than there are hardware queues: a graph whose DAG makes it reach for the third stream then crashes inside
# hipGraphLaunch (hip::Graph::UpdateStreams, seen with the reversible PHISeg3D backward graph at 128x128x64; 4 + 2 costs
# nothing on the PHiSeg graph, which happens to need two).  Keep the two numbers consistent.  Measured pairs (hw queues, graph
# streams), PHiSeg step / with the overlapped data-parallel exchange: (2,2) 20.8 / 21.4 ms, (3,3) 20.2 / 21.1, (4,4) 20.7 / 34.8,
# (2,4) 20.2 / 21.4 but unsafe -> 3 and 3.
try:
    _os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", str(max(1, min(4, int(_os.environ["GPU_MAX_HW_QUEUES"])))))
except ValueError:
    pass
