"""mgnet_amd -- MI355X-native implementation of MGNet's training hot path.

Only what the path needs lives here: `csrc/` (HIP kernels + the C-ABI of include/mgnet_hip.h), `_C` (the ctypes
binding of that ABI) and the host-side mirror of the reference's interface for the path (`modeling`, `geometry`).
There is no CPU fallback: importing an op without the built extension raises.
"""
__version__ = "0.1.0"

from .config import add_mgnet_config, get_cfg  # noqa: E402,F401
from . import modeling  # noqa: E402,F401  (registers MGNet, the heads and the backbone builder)

__all__ = ["add_mgnet_config", "get_cfg", "modeling"]
