"""raymarcher_amd — MI355X-native sphere-tracing renderer (HIP kernels behind a C-ABI).

Python here is plumbing: ctypes bindings of include/raymarcher_amd.h, torch for device memory /
streams / torch.distributed.  The renderer itself is raymarcher_amd/lib/libraymarcher_amd.so.
"""
from . import abi  # noqa: F401
from ._lib import LIB_PATH, RaymarcherError, lib  # noqa: F401
from .render import Renderer, Scene, SceneTables  # noqa: F401
