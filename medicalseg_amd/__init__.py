"""medicalseg_amd -- MI355X-native VNet hot path with medicalseg's API surface.

Compute is exclusively libmsegk.so (HIP, gfx950) through ctypes; importing the package is
cheap and does not need a GPU, creating the device (first model / tensor) does."""
from . import _lib

_lib.load()  # raises early when libmsegk.so has not been built: there is no fallback path
from . import nn, optimizer, parallel  # noqa: F401
from .cvlibs import Config, manager  # noqa: F401
from . import models, datasets, transforms  # noqa: F401  (populate the registries)
from .device import Tensor, IntTensor, get_device, to_tensor  # noqa: F401

__version__ = "0.1.0"
