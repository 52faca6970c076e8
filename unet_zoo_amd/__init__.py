"""Import shim: the package directory is named ``unet-zoo_amd`` (not a valid Python identifier),
so ``import unet_zoo_amd`` resolves here and re-points the package path at that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "unet-zoo_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
