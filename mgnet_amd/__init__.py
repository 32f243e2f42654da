"""mgnet_amd -- MI355X-native implementation of MGNet's training hot path.

Only what the path needs lives here: `csrc/` (HIP kernels + the C-ABI of include/mgnet_hip.h), `_C` (the ctypes
binding of that ABI) and the host-side mirror of the reference's interface for the path (`modeling`, `geometry`).
There is no CPU fallback: importing an op without the built extension raises.
"""
__version__ = "0.1.0"
