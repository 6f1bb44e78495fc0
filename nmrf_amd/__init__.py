"""nmrf_amd: MI355X-native (gfx950) implementation of the NMRF-Stereo inference hot path.

Host side mirrors the reference's `nmrf.models` / `ops.functions` interface; all hot-path arithmetic
is in libnmrf_hip.so (include/nmrf_hip.h), reached through ctypes.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
